"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the classifier path used by finetuning (`backbone(freq_x, class_head=True)`): per-modality features, modality
fusion, class layer, cross-entropy.  State dict in, tensors out, as in the rest of oracle/."""
import torch
import torch.nn.functional as F

from .deepsense import deepsense_forward
from .swt import swt_forward


def transformer_fusion(P, prefix, x, num_heads):
    """TransformerFusionBlock.forward without rescale factors, models/FusionModules.py:122-140: x [b, i, s, c] ->
    LayerNorm, query = mean over the s fused tokens, nn.MultiheadAttention(query, x, x) -> [b, i, c]."""
    b, i, s, c = x.shape
    x = x.reshape(b * i, s, c)
    x = F.layer_norm(x, (c,), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], 1e-5)
    q = x.mean(dim=1, keepdim=True)
    out, _ = F.multi_head_attention_forward(
        q.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1), c, num_heads,
        P[f"{prefix}.mha.in_proj_weight"], P[f"{prefix}.mha.in_proj_bias"], None, None, False, 0.0,
        P[f"{prefix}.mha.out_proj.weight"], P[f"{prefix}.mha.out_proj.bias"], training=False, need_weights=False)
    return out.transpose(0, 1).reshape(b, i, c)


def class_layer(P, x):
    """nn.Sequential(Linear) or (Linear, GELU, Linear) -- models/SW_Transformer.py:163-178, models/DeepSense.py:93-104."""
    x = F.linear(x, P["class_layer.0.weight"], P["class_layer.0.bias"])
    if "class_layer.2.weight" in P:
        x = F.linear(F.gelu(x), P["class_layer.2.weight"], P["class_layer.2.bias"])
    return x


def classifier_logits(model, P, cfg, freq_x, train=False, new_buffers=None):
    """`backbone(freq_x, class_head=True)`: SW_Transformer.py:244-276 (stack -> [b, 1, M, c] -> mod_fusion_layers -> class layer),
    DeepSense.py:139-157 (concatenate -> class layer)."""
    mods = cfg["modality_names"]
    if model == "SW_Transformer":
        feats = swt_forward(P, cfg, freq_x, proj_head=False)
        x = torch.stack([feats[m] for m in mods], dim=1).unsqueeze(1)  # [b, 1, M, c]
        fused = transformer_fusion(P, "mod_fusion_layers", x, cfg["SW_Transformer"]["loc_head_num"]).flatten(start_dim=1)
        return class_layer(P, fused)
    if model == "DeepSense":
        feats = deepsense_forward(P, cfg, freq_x, proj_head=False, train=train, new_buffers=new_buffers)
        return class_layer(P, torch.cat([feats[m] for m in mods], dim=1))
    raise Exception(f"Invalid model provided: {model}")


def finetune_param_filter(name):
    """general_utils/weight_utils.py:61-80 (FOCAL): only the class layer and the modality fusion layer are trained."""
    return "class_layer" in name or "mod_fusion_layer" in name


def finetune_loss_and_grads(model, P, cfg, freq_x, labels, train=False):
    """CrossEntropyLoss(logits, labels) and its gradient w.r.t. the learnable (finetune) parameters."""
    P = dict(P)
    keys = [k for k, v in P.items() if v.is_floating_point() and finetune_param_filter(k)]
    for k in keys:
        P[k] = P[k].detach().clone().requires_grad_(True)
    logits = classifier_logits(model, P, cfg, freq_x, train=train, new_buffers={})
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, [P[k] for k in keys], allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(keys, grads))


def supervised_param_filter(name):
    """Parameters that receive a gradient when the whole classifier is trained (train_utils/supervised_train.py:37 hands every
    parameter to the optimizer; `backbone(freq_x, class_head=True)` never touches the contrastive projectors, the unused
    absolute position embedding or DeepSense's single-location extras, whose `grad` stays None)."""
    dead = ("absolute_pos_embed.", "mod_extractors.", "loc_fusion_layers.", "loc_context_layers.", "loc_fusion_layer.", "mod_projectors.")
    return not name.startswith(dead)


def supervised_loss_and_grads(model, P, cfg, freq_x, labels, train=True):
    """One supervised step's loss and the gradient of EVERY trained parameter (patch embedding included)."""
    P = dict(P)
    keys = [k for k, v in P.items() if v.is_floating_point() and supervised_param_filter(k)
            and not k.endswith(("running_mean", "running_var", "attn_mask"))]
    for k in keys:
        P[k] = P[k].detach().clone().requires_grad_(True)
    logits = classifier_logits(model, P, cfg, freq_x, train=train, new_buffers={})
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, [P[k] for k in keys], allow_unused=True)
    return logits.detach(), loss.detach(), dict(zip(keys, grads))
