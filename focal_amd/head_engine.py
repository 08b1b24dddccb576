"""Classifier head of the finetuning path (`backbone(freq_x, class_head=True)`), as one engine object for backbone.StageFn.

SW_Transformer (models/SW_Transformer.py:269-276): stack the per-modality features [B, M, E] -> TransformerFusionBlock
(models/FusionModules.py:61-140: LayerNorm, query = mean over the M tokens, nn.MultiheadAttention over them) -> class layer.
DeepSense (models/DeepSense.py:154-157): concatenate the features -> class layer.
Finetuning trains exactly these parameters (general_utils/weight_utils.py:61-80); the encoders run forward-only in front.
Everything here is fp32 (activations of a few hundred KB); the products run on the GEMM family with `compute` operands."""
import torch

from . import ops
from ._lib import ACT_NONE, EPI_GELU, EPI_NONE


class ClassifierHead:
    def __init__(self, backbone, fusion_prefix=None, heads=0, p_attn=0.0):
        self.bb, self.fusion, self.heads, self.p_attn = backbone, fusion_prefix, heads, p_attn

    # ------------------------------------------------------------------------------------------------ helpers
    def _lin(self, x, w, b, act_in=ACT_NONE, epi=EPI_NONE, act_grad=None):
        ar, cc, f32 = self.bb.arena(), ops.code(self.bb.compute_dtype), ops.code(torch.float32)
        M, K = x.shape
        N = ar.index[w][2][0]
        d = ops.linear_desc(cc, M, N, K, f32, f32, act_in, epi)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ops.linear_fwd(d, x, ar.operand(w), ar.master(b), None, y, act_grad)
        return y, d

    def _lin_bwd(self, d, dy, x, w, b, need_dx=True, aux=None):
        ar = self.bb.arena()
        ops.linear_bwd_weight(d, dy, x, ar.g(w), ar.g(b))
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        ops.linear_bwd_data(d, dy, ar.operand(w), aux, dx)
        return dx

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, feats, training):
        """feats: [B, M, E] (fusion) or [B, sum E_m] (concatenation) fp32 -> logits [B, n_cls]."""
        bb, ar = self.bb, self.bb.arena()
        sv = {"training": training}
        if self.fusion is not None:
            B, M, E = feats.shape
            pre = self.fusion
            x = feats.reshape(B * M, E).contiguous()
            xn, st = ops.layernorm_fwd(x, ar.master(f"{pre}.norm1.weight"), ar.master(f"{pre}.norm1.bias"), torch.float32)
            qin = ops.mean_time(xn, B, M, E)
            # nn.MultiheadAttention packs W_q | W_k | W_v as in_proj_weight [3E, E]: views of the arena, no copies
            wq, wkv = _ProjView(ar, f"{pre}.mha.in_proj_weight", 0, E), _ProjView(ar, f"{pre}.mha.in_proj_weight", E, 3 * E)
            bq, bkv = _ProjView(ar, f"{pre}.mha.in_proj_bias", 0, E), _ProjView(ar, f"{pre}.mha.in_proj_bias", E, 3 * E)
            q = self._proj(qin, wq, bq)
            kv = self._proj(xn, wkv, bkv)
            o = torch.empty(B, E, dtype=torch.float32, device=x.device)
            probs = torch.empty(B, self.heads, M, dtype=torch.float32, device=x.device)
            weights = torch.empty_like(probs)
            p = self.p_attn if training else 0.0
            ops.fusion_attn_fwd(B, M, E, self.heads, q, kv, o, probs, weights, bb.rng_state() if p > 0 else None, 0x7F00, p)
            y, d_out = self._lin(o, f"{pre}.mha.out_proj.weight", f"{pre}.mha.out_proj.bias")
            sv.update(x=x, st=st, xn=xn, qin=qin, q=q, kv=kv, o=o, probs=probs, weights=weights, d_out=d_out, B=B, M=M, E=E)
            h = y
        else:
            h = feats.contiguous()
        sv["h_in"] = h
        if "class_layer.2.weight" in ar.index:
            # Linear -> GELU -> Linear (`pretrained_head` != "linear", models/SW_Transformer.py:175-181): the hidden layer on the exact-fp32
            # GEMM with the GELU epilogue (value and derivative from one launch), the few-column output layer as for the linear head
            f32 = ops.code(torch.float32)
            Bn, K = h.shape
            N = ar.index["class_layer.0.weight"][2][0]
            d0 = ops.linear_desc(f32, Bn, N, K, f32, f32, ACT_NONE, EPI_GELU)
            h1, g1 = torch.empty(Bn, N, dtype=torch.float32, device=h.device), torch.empty(Bn, N, dtype=torch.float32, device=h.device)
            ops.linear_fwd(d0, h, ar.master("class_layer.0.weight"), ar.master("class_layer.0.bias"), None, h1, g1)
            sv.update(d0=d0, h1=h1, g1=g1)
            logits = ops.small_linear_fwd(h1, ar.master("class_layer.2.weight"), ar.master("class_layer.2.bias"))
            return logits, sv
        logits = ops.small_linear_fwd(h, ar.master("class_layer.0.weight"), ar.master("class_layer.0.bias"))
        return logits, sv

    def _proj(self, x, w, b):
        cc, f32 = ops.code(self.bb.compute_dtype), ops.code(torch.float32)
        d = ops.linear_desc(cc, x.shape[0], w.rows, x.shape[1], f32, f32)
        y = torch.empty(x.shape[0], w.rows, dtype=torch.float32, device=x.device)
        ops.linear_fwd(d, x, w.operand(), b.master(), None, y)
        return y

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, sv, dlogits):
        ar = self.bb.arena()
        if dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
            dlogits = dlogits.float().contiguous()
        train_encoders = getattr(self.bb, "supervised", False)  # supervised training: the gradient continues into the encoders
        need_dx = self.fusion is not None or train_encoders
        if "d0" in sv:
            dh1 = ops.small_linear_bwd(dlogits, sv["h1"], ar.master("class_layer.2.weight"), ar.g("class_layer.2.weight"),
                                       ar.g("class_layer.2.bias"), True)
            ops.mul_(dh1, sv["g1"])  # through the GELU: the derivative saved by the forward epilogue
            ops.linear_bwd_weight(sv["d0"], dh1, sv["h_in"], ar.g("class_layer.0.weight"), ar.g("class_layer.0.bias"))
            dh = None
            if need_dx:
                dh = torch.empty_like(sv["h_in"])
                d0b = ops.linear_desc(sv["d0"].dtype, sv["d0"].M, sv["d0"].N, sv["d0"].K, sv["d0"].x_dtype, sv["d0"].y_dtype)
                ops.linear_bwd_data(d0b, dh1, ar.master("class_layer.0.weight"), None, dh)
        else:
            dh = ops.small_linear_bwd(dlogits, sv["h_in"], ar.master("class_layer.0.weight"), ar.g("class_layer.0.weight"),
                                      ar.g("class_layer.0.bias"), need_dx)
        if self.fusion is None:
            return dh if train_encoders else None  # finetuning: the encoders in front are frozen
        pre, B, M, E = self.fusion, sv["B"], sv["M"], sv["E"]
        do = self._lin_bwd(sv["d_out"], dh, sv["o"], f"{pre}.mha.out_proj.weight", f"{pre}.mha.out_proj.bias")
        dq = torch.empty(B, E, dtype=torch.float32, device=do.device)
        dkv = torch.empty(B * M, 2 * E, dtype=torch.float32, device=do.device)
        ops.fusion_attn_bwd(B, M, E, self.heads, sv["q"], sv["kv"], sv["probs"], sv["weights"], do, dq, dkv)
        wq, wkv = _ProjView(ar, f"{pre}.mha.in_proj_weight", 0, E), _ProjView(ar, f"{pre}.mha.in_proj_weight", E, 3 * E)
        bq, bkv = _ProjView(ar, f"{pre}.mha.in_proj_bias", 0, E), _ProjView(ar, f"{pre}.mha.in_proj_bias", E, 3 * E)
        dqin = self._proj_bwd(dq, sv["qin"], wq, bq)
        dxn = self._proj_bwd(dkv, sv["xn"], wkv, bkv)
        dxn.view(B, M, E).add_(dqin.view(B, 1, E) / M)  # the query is the mean of the M normalised tokens
        dx = torch.empty_like(sv["x"])                  # gradient w.r.t. the features (propagated further in supervised training only)
        ops.layernorm_bwd(dxn, sv["x"], sv["st"], ar.master(f"{pre}.norm1.weight"), dx, False,
                          ar.g(f"{pre}.norm1.weight"), ar.g(f"{pre}.norm1.bias"))
        return dx.view(B, M, E) if train_encoders else None

    def _proj_bwd(self, dy, x, w, b):
        cc, f32 = ops.code(self.bb.compute_dtype), ops.code(torch.float32)
        d = ops.linear_desc(cc, x.shape[0], w.rows, x.shape[1], f32, f32)
        ops.linear_bwd_weight(d, dy, x, w.grad(), b.grad())
        dx = torch.empty_like(x)
        ops.linear_bwd_data(d, dy, w.operand(), None, dx)
        return dx


class _ProjView:
    """Rows [lo, hi) of a packed projection parameter (nn.MultiheadAttention's in_proj_weight / in_proj_bias) as GEMM operands."""

    def __init__(self, arena, name, lo, hi):
        self.ar, self.name, self.lo, self.hi, self.rows = arena, name, lo, hi, hi - lo

    def operand(self):
        return self.ar.operand(self.name)[self.lo:self.hi]

    def master(self):
        return self.ar.master(self.name)[self.lo:self.hi]

    def grad(self):
        return self.ar.g(self.name)[self.lo:self.hi]
