"""Execution of one (location, modality) SW_Transformer encoder on the HIP kernels, forward and backward.

The reference runs ~35 ATen ops per Swin block through autograd (models/SwinModules.py:294-343).  Here a block is
7 kernels forward / 13 backward, launched back to back on the current stream with explicitly managed saved
activations; torch.autograd only sees one node per encoder (see `EncoderFn` in models/SW_Transformer.py).

Per block (M = B*H*W tokens, C channels, CT = matrix-core operand dtype):
  forward   a1,st1 = LN(x) . qkv = a1 Wqkv^T + b . o = W-MSA(qkv) . x_mid = x + drop(o Wp^T + b)
            a2,st2 = LN(x_mid) . (h, h') = drop(gelu(a2 W1^T + b)) and its derivative . x_out = x_mid + drop(h W2^T + b)
  saved     x, st1, a1, qkv, o, x_mid, st2, a2, h, h'   (x fp32; the rest in CT; dropout masks are regenerated)
  backward  the residual-gradient stream g (fp32 [M, C]) is updated in place through the block.
"""
import os

import torch

from . import ops
from ._lib import ACT_GELU, ACT_NONE, EPI_GELU, EPI_NONE, EPI_RESIDUAL


def tail_fp32(bb, which="mod_in"):
    """The projector (two [2B, E] x [E, E] products per modality) multiplies fp32 operands on the fp32 matrix-core path even in bf16 mode
    (round 6): it sees the accumulated rounding of the whole encoder once more, and at M = 2B rows fp32 costs nothing
    (step: +-0.3 %, inside the box's noise).  B = 8 fixture, embeddings max |d| / max |ref| (audio / seismic) and the ranking term's error:
        FOCAL_TAIL_FP32=none (round 5)   0.960e-2 / 0.892e-2   rank 0.984e-2
        projector (default)              0.835e-2 / 0.838e-2   rank 0.973e-2
        mod_in                           1.007e-2 / 0.892e-2   (fails)
        both                             0.789e-2 / 0.837e-2   rank 1.020e-2 (fails)
    -- the statistic is a maximum over 2 048 entries / a two-element hinge and moves chaotically with WHICH operands are rounded (the CPU
    emulation of the same four forms, tests/bf16_error_attribution.py: 0.98 / 0.91 / 0.89 / 0.84e-2 and rank 1.48 / 0.96 / 1.33 / 0.95e-2):
    every site contributes 5-25 % of a random walk, no cheap site dominates, and a robust 0.8e-2 needs fp32 (or hi + lo split) operands in
    the stage-1/2 blocks themselves (profiles/r6_bf16_attribution.txt).  mod_in_layers can be added with FOCAL_TAIL_FP32=both."""
    if bb.compute_dtype == torch.float32:
        return False
    sel = os.environ.get("FOCAL_TAIL_FP32", "projector")
    return sel == "both" or sel == which


def tail_weight(bb, ar, name, which="mod_in"):
    return ar.master(name) if tail_fp32(bb, which) else ar.operand(name)


class SwinModEncoder:
    def __init__(self, backbone, loc, mod, mod_index):
        self.bb = backbone
        self.loc, self.mod, self.mod_index = loc, mod, mod_index
        self.geo = backbone.geometry[loc][mod]
        self.pre = f"freq_interval_layers.{loc}.{mod}"

    # ------------------------------------------------------------------ helpers
    def _drop(self, rng, view, uid, site, p_elem, p_path, rows_per_sample):
        base = ((view * 8 + self.mod_index) * 64 + uid) * 8
        return ops.drop_desc(rng, base + site, p_elem, base + site + 4, p_path, rows_per_sample)

    def forward(self, x_freq, view, training):
        """x_freq: [B, c, i, s] fp32 -> (feat [B, loc_out] fp32, saved state for backward)."""
        bb, geo, ar = self.bb, self.geo, self.bb.arena()
        ct = bb.compute_dtype
        cc = ops.code(ct)
        f32 = ops.code(torch.float32)
        rng = bb.rng_state() if training else None
        B = x_freq.shape[0]
        if x_freq.dtype != torch.float32 or not x_freq.is_contiguous():
            x_freq = x_freq.float().contiguous()
        pe = f"patch_embed.{self.loc}.{self.mod}"
        P = bb.param  # cold parameters (frozen patch embedding) are read where they live
        # LayerNorm forwards ride on their producers (fuse_ln) and the 64-channel MLP branch is one kernel (fuse_mlp) wherever the library
        # has the kernel for the dtype / shape (ops.*_supported; the exact-fp32 mode runs the plain kernels).
        # The LayerNorm epilogue of the residual GEMMs beyond 64 channels (row-complete wave tiles on the LDS-DMA GEMM): at 128 channels
        # (+1.1 % on the step, same-box), not at 256, where one row fragment per wave makes the fused fc2 take 55-61 us against 35 + 11
        # (-1 % with both on; profiles/r3_ln_wide_ab.txt).
        fuse_ln = fuse_mlp = True
        LN_FUSE_MAX_C = 128
        pre_ln = None  # (a1, st1) of the next block when the kernel before it already produced them
        first = f"{self.pre}.0.blocks.0"
        embed_saved = None
        if getattr(bb, "supervised", False):
            # Supervised training from scratch trains the patch embedding too (it is frozen only in FOCAL pretraining): the strided
            # Conv2d runs on the patchifying convolution kernels that already have a weight gradient (focal_conv_in_*: DeepSense's first
            # layer is the same operator), on the zero-padded spectrum; its LayerNorm on the LayerNorm kernels.  The fused
            # pad + embed + LayerNorm kernel of the pretraining path has no backward.
            Hp, Wp, pw = geo["grid"][0], geo["grid"][1], geo["patch"][1]
            _, cin, I, S = x_freq.shape
            xpad = torch.nn.functional.pad(x_freq, (0, Wp * pw - S, 0, Hp - I))  # data movement only
            d_pe = ops.conv_in_desc(B, cin, Hp, Wp * pw, Wp, pw, pw, 0, geo["stages"][0]["C"])
            z = ops.conv_in_fwd(d_pe, xpad, ar.master(f"{pe}.proj.weight"), ar.master(f"{pe}.proj.bias"))
            x, st_e = ops.layernorm_fwd(z, ar.master(f"{pe}.norm.weight"), ar.master(f"{pe}.norm.bias"), torch.float32)
            embed_saved = dict(d=d_pe, xpad=xpad, z=z, st=st_e)
        elif fuse_ln and geo["stages"][0]["C"] == 64:  # the embedding kernel also emits block 0's norm1
            x, a1_0, st1_0 = ops.pad_patch_embed_ln(x_freq, P(f"{pe}.proj.weight"), P(f"{pe}.proj.bias"), P(f"{pe}.norm.weight"),
                                                    P(f"{pe}.norm.bias"), geo["grid"][0], geo["grid"][1], geo["patch"][1],
                                                    next_ln=(ar.master(f"{first}.norm1.weight"), ar.master(f"{first}.norm1.bias"), ct))
            pre_ln = (a1_0, st1_0)
        else:
            x = ops.pad_patch_embed_ln(x_freq, P(f"{pe}.proj.weight"), P(f"{pe}.proj.bias"), P(f"{pe}.norm.weight"),
                                       P(f"{pe}.norm.bias"), geo["grid"][0], geo["grid"][1], geo["patch"][1])
        if bb.config["APE"]:
            raise ops._lib.FocalHipError("absolute position embedding (APE: True) is outside the HIP hot path")
        p_drop = bb.drop_rate if training else 0.0
        p_attn = bb.attn_drop_rate if training else 0.0
        saved = {"B": B, "view": view, "training": training, "blocks": [], "merges": [], "embed": embed_saved}
        uid = 0
        # Where a GEMM wave owns whole rows (64 channels: the 64 x 64 tile; 128 / 256 channels in bf16: the LDS-DMA kernel with
        # row-complete wave tiles) the residual GEMMs also emit the LayerNorm that follows them (norm2 after proj, the next
        # block's norm1 after fc2): no separate pass over the residual stream for those LayerNorms.
        for si, st in enumerate(geo["stages"]):
            H, W, Cc = st["H"], st["W"], st["C"]
            L = H * W
            M = B * L
            for bi in range(st["depth"]):
                pb = f"{self.pre}.{si}.blocks.{bi}"
                wh, ww, sh, sw = bb.block_windows[(self.loc, self.mod, si, bi)]
                p_path = bb.drop_path_rates[self.mod][uid] if training else 0.0
                if pre_ln is not None:
                    a1, st1 = pre_ln
                    pre_ln = None
                else:
                    a1, st1 = ops.layernorm_fwd(x, ar.master(f"{pb}.norm1.weight"), ar.master(f"{pb}.norm1.bias"), ct)
                d_qkv = ops.linear_desc(cc, M, 3 * Cc, Cc, cc, cc)
                d_att = ops.attn_desc(cc, B, H, W, Cc, geo["heads"], wh, ww, sh, sw, p_attn, rng,
                                      ((view * 8 + self.mod_index) * 64 + uid) * 8 + 3)
                o = torch.empty(M, Cc, dtype=ct, device=x.device)
                if ops.attn_qkv_supported(cc, Cc, geo["heads"], wh * ww):
                    # 64-channel blocks: the attention kernel projects q / k / v of a (window, head) item itself -- no qkv GEMM launch,
                    # no [M, 3C] tensor written, read back or saved (the backward kernel recomputes it from a1)
                    qkv = None
                    ops.window_attn_qkv_fwd(d_att, a1, ar.operand(f"{pb}.attn.qkv.weight"), ar.master(f"{pb}.attn.qkv.bias"),
                                            ar.master(f"{pb}.attn.relative_position_bias_table"), o)
                else:
                    qkv = torch.empty(M, 3 * Cc, dtype=ct, device=x.device)
                    ops.linear_fwd(d_qkv, a1, ar.operand(f"{pb}.attn.qkv.weight"), ar.master(f"{pb}.attn.qkv.bias"), None, qkv)
                    ops.window_attn_fwd(d_att, qkv, ar.master(f"{pb}.attn.relative_position_bias_table"), o)
                d_proj = ops.linear_desc(cc, M, Cc, Cc, cc, f32, ACT_NONE, EPI_RESIDUAL,
                                         out_drop=self._drop(rng, view, uid, 0, p_drop, p_path, L))
                x_mid = torch.empty(M, Cc, dtype=torch.float32, device=x.device)
                fuse_wide = Cc <= LN_FUSE_MAX_C
                # 64-channel blocks (round 6): proj + residual + norm2 ride in FRONT of the fused MLP, in its launch (ops.mlp_proj_fwd below)
                proj_in_mlp = fuse_ln and fuse_mlp and ops.mlp_supported(ct, Cc, 4 * Cc) and ops.mlp_proj_supported(ct, Cc, 4 * Cc)
                # ... and so do the 128 / 256-channel ones in front of the one-launch MLP of stages 1-2 (ops.mlp_wide_proj_fwd)
                proj_in_wide = fuse_ln and fuse_mlp and ops.mlp_wide_supported(ct, Cc, 4 * Cc) and ops.mlp_wide_proj_supported(ct, Cc, 4 * Cc)
                if proj_in_mlp or proj_in_wide:
                    a2 = st2 = None
                elif fuse_ln and (Cc == 64 or fuse_wide) and ops.resid_ln_supported(cc, Cc, Cc):
                    a2, st2 = ops.linear_resid_ln_fwd(d_proj, o, ar.operand(f"{pb}.attn.proj.weight"), ar.master(f"{pb}.attn.proj.bias"),
                                                      x, x_mid, ar.master(f"{pb}.norm2.weight"), ar.master(f"{pb}.norm2.bias"), ct)
                else:
                    ops.linear_fwd(d_proj, o, ar.operand(f"{pb}.attn.proj.weight"), ar.master(f"{pb}.attn.proj.bias"), x, x_mid)
                    a2, st2 = ops.layernorm_fwd(x_mid, ar.master(f"{pb}.norm2.weight"), ar.master(f"{pb}.norm2.bias"), ct)
                d_fc1 = ops.linear_desc(cc, M, 4 * Cc, Cc, cc, cc, ACT_NONE, EPI_GELU,
                                        out_drop=self._drop(rng, view, uid, 1, p_drop, 0.0, L))
                d_fc2 = ops.linear_desc(cc, M, Cc, 4 * Cc, cc, f32, ACT_GELU, EPI_RESIDUAL,
                                        out_drop=self._drop(rng, view, uid, 2, p_drop, p_path, L))
                x_out = torch.empty(M, Cc, dtype=torch.float32, device=x.device)
                d_mlp = h = hg = None
                if fuse_mlp and ops.mlp_supported(ct, Cc, 4 * Cc):
                    # the whole MLP branch in one kernel: the [M, 4C] hidden activation never reaches HBM and is recomputed in backward
                    d_mlp = ops.mlp_desc(cc, M, Cc, 4 * Cc, d_fc1.out_drop, d_fc2.out_drop)
                    nxt_ln = None
                    if fuse_ln and bi + 1 < st["depth"]:
                        nb = f"{self.pre}.{si}.blocks.{bi + 1}"
                        nxt_ln = (ar.master(f"{nb}.norm1.weight"), ar.master(f"{nb}.norm1.bias"))
                    mlp_bits = ops.mlp_mask_bits(d_mlp, x.device)  # the hidden dropout's keep bits (32 B per token): all the branch saves beside a2
                    if proj_in_mlp:
                        (a2, st2), pre_ln = ops.mlp_proj_fwd(d_mlp, o, x, ar.operand(f"{pb}.attn.proj.weight"), ar.master(f"{pb}.attn.proj.bias"), d_proj.out_drop,
                                                             ar.master(f"{pb}.norm2.weight"), ar.master(f"{pb}.norm2.bias"), x_mid,
                                                             ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"),
                                                             ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"), x_out, next_ln=nxt_ln, mask_bits=mlp_bits)
                    else:
                        pre_ln = ops.mlp_fwd(d_mlp, a2, x_mid, ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"),
                                             ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"), x_out, next_ln=nxt_ln, mask_bits=mlp_bits)
                    saved["blocks"].append(dict(pb=pb, x=x, st1=st1, a1=a1, qkv=qkv, o=o, x_mid=x_mid, st2=st2, a2=a2, h=None, hg=None, mlp_bits=mlp_bits,
                                                d_qkv=d_qkv, d_att=d_att, d_proj=d_proj, d_fc1=d_fc1, d_fc2=d_fc2, d_mlp=d_mlp, M=M, C=Cc))
                    x = x_out
                    uid += 1
                    continue
                h = torch.empty(M, 4 * Cc, dtype=ct, device=x.device)
                hg = torch.empty(M, 4 * Cc, dtype=ct, device=x.device)  # d h / d(pre-activation), dropout included
                want_ln = fuse_ln and (Cc == 64 or fuse_wide) and ops.resid_ln_supported(cc, Cc, 4 * Cc) and bi + 1 < st["depth"]
                if fuse_mlp and ops.mlp_wide_supported(ct, Cc, 4 * Cc):
                    # stages 1-2 (round 6): fc1 -> GELU -> dropout -> fc2 -> residual (+ the next LayerNorm at 128 channels) as ONE launch; h and
                    # hg are written once for the backward pass and never read back (bit-identical to the two launches below)
                    d_wide = ops.mlp_desc(cc, M, Cc, 4 * Cc, d_fc1.out_drop, d_fc2.out_drop)
                    nxt_ln = None
                    if want_ln or (fuse_ln and bi + 1 < st["depth"] and os.environ.get("FOCAL_MLP_WIDE_LN256") != "0"):
                        # (256 channels too: a wave of this kernel owns whole rows, and its LayerNorm repeats ln_fwd_kernel's summation tree --
                        # the next block's norm1 comes out bit-identical to the stand-alone launch it replaces)
                        nb = f"{self.pre}.{si}.blocks.{bi + 1}"
                        nxt_ln = (ar.master(f"{nb}.norm1.weight"), ar.master(f"{nb}.norm1.bias"))
                    if proj_in_wide:
                        (a2, st2), pre_ln = ops.mlp_wide_proj_fwd(d_wide, o, x, ar.operand(f"{pb}.attn.proj.weight"), ar.master(f"{pb}.attn.proj.bias"), d_proj.out_drop,
                                                                  ar.master(f"{pb}.norm2.weight"), ar.master(f"{pb}.norm2.bias"), x_mid,
                                                                  ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"),
                                                                  ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"), x_out, h, hg, next_ln=nxt_ln)
                    else:
                        pre_ln = ops.mlp_wide_fwd(d_wide, a2, x_mid, ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"),
                                                  ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"), x_out, h, hg, next_ln=nxt_ln)
                    saved["blocks"].append(dict(pb=pb, x=x, st1=st1, a1=a1, qkv=qkv, o=o, x_mid=x_mid, st2=st2, a2=a2, h=h, hg=hg,
                                                d_qkv=d_qkv, d_att=d_att, d_proj=d_proj, d_fc1=d_fc1, d_fc2=d_fc2, M=M, C=Cc))
                    x = x_out
                    uid += 1
                    continue
                ops.linear_fwd(d_fc1, a2, ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"), None, h, hg)
                if fuse_ln and (Cc == 64 or fuse_wide) and ops.resid_ln_supported(cc, Cc, 4 * Cc) and bi + 1 < st["depth"]:
                    nb = f"{self.pre}.{si}.blocks.{bi + 1}"
                    pre_ln = ops.linear_resid_ln_fwd(d_fc2, h, ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"),
                                                     x_mid, x_out, ar.master(f"{nb}.norm1.weight"), ar.master(f"{nb}.norm1.bias"), ct)
                else:
                    ops.linear_fwd(d_fc2, h, ar.operand(f"{pb}.mlp.fc2.weight"), ar.master(f"{pb}.mlp.fc2.bias"), x_mid, x_out)
                saved["blocks"].append(dict(pb=pb, x=x, st1=st1, a1=a1, qkv=qkv, o=o, x_mid=x_mid, st2=st2, a2=a2, h=h, hg=hg,
                                            d_qkv=d_qkv, d_att=d_att, d_proj=d_proj, d_fc1=d_fc1, d_fc2=d_fc2,
                                            M=M, C=Cc))
                x = x_out
                uid += 1
            if st["downsample"]:
                pm = f"{self.pre}.{si}.downsample"
                gather = (B, H, W, Cc)
                a4, st4 = ops.layernorm_fwd(x, ar.master(f"{pm}.norm.weight"), ar.master(f"{pm}.norm.bias"), ct, gather=gather)
                M4 = M // 4
                d_red = ops.linear_desc(cc, M4, 2 * Cc, 4 * Cc, cc, f32)
                x_next = torch.empty(M4, 2 * Cc, dtype=torch.float32, device=x.device)
                ops.linear_fwd(d_red, a4, ar.operand(f"{pm}.reduction.weight"), None, None, x_next)
                saved["merges"].append(dict(pm=pm, x=x, st4=st4, a4=a4, d_red=d_red, gather=gather, after_block=len(saved["blocks"])))
                x = x_next
        last = geo["stages"][-1]
        K = last["H"] * last["W"] * last["C"]
        pin = f"mod_in_layers.{self.loc}.{self.mod}"
        n_out = bb.config["loc_out_channels"]
        splits = max(1, min(32, K // 512))
        d_in = ops.linear_desc(f32 if tail_fp32(bb) else cc, B, n_out, K, f32, f32, splits=splits)
        feat = torch.zeros(B, n_out, dtype=torch.float32, device=x.device)
        ops.linear_fwd(d_in, x, tail_weight(bb, ar, f"{pin}.weight"), ar.master(f"{pin}.bias"), None, feat)
        saved.update(x_final=x, d_in=d_in, pin=pin)
        return feat, saved

    def _mlp_partials(self, d_mlp, dev):
        """Workspace of the fused MLP backward's weight-gradient flush (one per encoder: its blocks run one after the other on the
        encoder's stream); FOCAL_MLP_BWD_ATOMICS=1 keeps the round-4 atomic flush (same-box A/B)."""
        if os.environ.get("FOCAL_MLP_BWD_ATOMICS") == "1":
            return None
        # one workspace per size, never dropped: a captured step graph holds its address, and a smaller batch later (the last one of an
        # epoch) followed by the captured shape again must find the block where the graph left it
        pool = self.__dict__.setdefault("_mlp_ws", {})
        n = ops.mlp_bwd_partials_floats(d_mlp)
        ws = pool.get((n, dev))
        if ws is None:
            ws = pool[(n, dev)] = torch.empty(n, dtype=torch.float32, device=dev)
        return ws

    def backward(self, saved, dfeat):
        """Accumulates every parameter gradient of this encoder into the arena; returns nothing (input is a leaf).

        Data parallel (bb.split_backward, set by focal_amd/graph_step.py): the pass stops once the LAST stage's blocks are done -- by
        then the gradients of that stage, of mod_in and of the projector (88 % of the arena's bytes) are final and their all-reduce can
        start -- and `backward_rest` runs the earlier stages beside that collective."""
        bb, ar = self.bb, self.bb.arena()
        ct = bb.compute_dtype
        dev = dfeat.device
        if dfeat.dtype != torch.float32 or not dfeat.is_contiguous():
            dfeat = dfeat.float().contiguous()
        pin, d_in, xf = saved["pin"], saved["d_in"], saved["x_final"]
        d_in_b = ops.linear_desc(d_in.dtype, d_in.M, d_in.N, d_in.K, d_in.x_dtype, d_in.y_dtype)  # no split on the way back
        ops.linear_bwd_weight(d_in_b, dfeat, xf, ar.g(f"{pin}.weight"), ar.g(f"{pin}.bias"))
        g = torch.empty_like(xf)
        ops.linear_bwd_data(d_in_b, dfeat, tail_weight(bb, ar, f"{pin}.weight"), None, g)
        blocks = saved["blocks"]
        last = blocks[-1]
        gm = ops.mask_cast(g.view(last["M"], last["C"]), last["d_fc2"].out_drop, ct)
        stop = 0
        if getattr(bb, "split_backward", False) and saved["merges"] and saved.get("embed") is None:
            stop = max(m["after_block"] for m in saved["merges"])  # first block of the last stage
        state = dict(g=g, gm=gm, next=len(blocks) - 1)
        self._backward_blocks(saved, state, stop)
        if stop > 0:
            saved["_resume"] = state
            bb.pending_backward.append((self, saved, torch.cuda.current_stream(dev)))
            return
        self._backward_tail(saved, state)

    def backward_rest(self, saved):
        """The earlier stages of a split backward pass (see backward)."""
        state = saved.pop("_resume")
        self._backward_blocks(saved, state, 0)
        self._backward_tail(saved, state)

    def _backward_blocks(self, saved, state, stop):
        """Blocks state["next"] .. stop, last first; state carries the residual-stream gradient g and its masked operand copy gm."""
        bb, ar = self.bb, self.bb.arena()
        ct = bb.compute_dtype
        g, gm = state["g"], state["gm"]
        dev = g.device
        merges = {m["after_block"]: m for m in saved["merges"]}
        blocks = saved["blocks"]
        cc = ops.code(ct)
        # gm = ct(g * mask): the residual-stream gradient times the dropout x drop-path mask of the branch that consumes it
        # next, written once by whichever kernel completes g (LayerNorm backward, or mask_cast for the first one); the
        # branch's dX / dW GEMMs then run as plain `ct` kernels instead of regenerating the mask per column tile.
        group_dw = True
        # LayerNorm backward as the epilogue of the dX GEMM in front of it (64 / 128 channels, bf16): focal_linear_bwd_data_ln; at 256
        # channels neither one nor two waves per row gained anything (profiles/r3_ln_bwd_fused_ab.txt)
        fuse_ln_bwd = ct == torch.bfloat16
        ln_bwd_max_c = int(os.environ.get("FOCAL_LN_BWD_MAX_C", "128"))  # (256: measured again in round 6, tools/ab_env.sh)
        # one pass per step over this encoder's weights (both views in one batch): a gradient tile has a single writer per launch
        exclusive_dw = bool(getattr(bb, "views_share_pass", False))
        # (A block's weight gradients on a side stream -- nothing in the backward pass waits for them -- was measured in round 3 and lost
        # 9 % / 20 %: their operands must then outlive the block, and every later temporary lands on memory the Infinity Cache does not
        # hold; profiles/r3_dw_stream_ab.txt.  The code was removed in round 4.)
        # FOCAL_DW_PAIR = n > 1: the group launches of n consecutive blocks (128 x 128 tiles: stages 1-2) go out as ONE launch.  A launch costs
        # ~8 us of ramp and ~8 us of memory-side fp32 atomics (one 64 KB tile per workgroup whatever the problem count: tools/lab_dwg.sh), and
        # twice the tiles need half the token slices -- but 96 tiles x 144 ring stages cut into <= 256 workgroups is 2 slices of 72 stages
        # (192 workgroups) against 5 of 29 (240), and the operands of the waiting block stay alive a block longer: measured -0.4 ... -0.9 % (2) and
        # -0.9 ... +0.7 % (4) inside the step (tools/ab_dw_pair.sh, profiles/r6_dw_group_fixed_cost.txt).  NOT the default.
        dw_pair = int(os.environ.get("FOCAL_DW_PAIR", "1"))
        pending = state.setdefault("dw_pending", [])

        def flush_dw():
            if pending:
                ops.linear_bwd_weight_group(cc, list(pending), exclusive=exclusive_dw)
                pending.clear()
                state["dw_pending_blocks"] = 0

        def submit_dw(items, mergeable):
            """Launch a block's grouped weight gradients, or park them for the next block's launch; True when they are still waiting."""
            if not mergeable or dw_pair <= 1:
                ops.linear_bwd_weight_group(cc, items, exclusive=exclusive_dw)
                return False
            pending.extend(items)
            state["dw_pending_blocks"] = state.get("dw_pending_blocks", 0) + 1
            if state["dw_pending_blocks"] >= dw_pair or len(pending) + 5 > ops.DW_GROUP_MAX_PROBLEMS:
                flush_dw()
                return False
            return True

        for k in range(state["next"], stop - 1, -1):
            if (k + 1) in merges:  # a PatchMerging sits between block k and block k+1
                mg = merges[k + 1]
                pm, d_red = mg["pm"], mg["d_red"]
                da4 = torch.empty_like(mg["a4"])
                if mg.get("dw_done"):  # the reduction's dW went out with block k+1's group; gm is the plain `ct` copy of g
                    d_red_b = ops.linear_desc(cc, d_red.M, d_red.N, d_red.K, cc, cc)
                    ops.linear_bwd_data(d_red_b, gm.view(d_red.M, d_red.N), ar.operand(f"{pm}.reduction.weight"), None, da4)
                else:
                    ops.linear_bwd_weight(d_red, g, mg["a4"], ar.g(f"{pm}.reduction.weight"), None)
                    ops.linear_bwd_data(d_red, g, ar.operand(f"{pm}.reduction.weight"), None, da4)
                g = torch.empty_like(mg["x"])
                gm = torch.empty(mg["x"].shape, dtype=ct, device=dev)
                ops.layernorm_bwd(da4, mg["x"], mg["st4"], ar.master(f"{pm}.norm.weight"), g, False,
                                  ar.g(f"{pm}.norm.weight"), ar.g(f"{pm}.norm.bias"), gather=mg["gather"],
                                  dx_masked=gm, mask=blocks[k]["d_fc2"].out_drop)
            s = blocks[k]
            pb, M, Cc = s["pb"], s["M"], s["C"]
            gm = gm.view(M, Cc)
            # The four weight gradients of a block (fc2, fc1, proj, qkv) as ONE launch on 128 x 128 tiles once all their operands
            # exist (ops.linear_bwd_weight_group; stages 1-2: C >= 128): one ramp and one atomic epilogue instead of four, half the
            # L2 -> LDS bytes per MAC.  Their operands stay alive until then, so the attention branch's masked gradient gets a buffer
            # of its own instead of overwriting the MLP branch's.
            # 64-channel blocks (stage 0) are too narrow for those tiles: their launches -- qkv and proj beside the fused MLP kernel, all four
            # otherwise -- go out as one launch of the 64 x 64 ring tiles behind a problem table (dw_group_kind 1).
            fused_mlp = s.get("d_mlp") is not None
            shapes = [(Cc, Cc), (3 * Cc, Cc)] + ([] if fused_mlp else [(Cc, 4 * Cc), (4 * Cc, Cc)])
            kinds = [ops.dw_group_kind(cc, M, n_, k_) for n_, k_ in shapes] if group_dw else [0]
            grouped = min(kinds) >= 1
            grouped128 = min(kinds) == 2
            dw_items, dw_calls = [], []  # grouped problems / single launches (descriptor, dy, x, dw, dbias) of this block

            def weight_grad(desc, dy, x, dw, db):
                if grouped:
                    dw_items.append((dy, x, dw, db))
                else:
                    ops.linear_bwd_weight(desc, dy, x, dw, db)
            # ---- MLP branch: x_out = x_mid + mask * (h W2^T + b2), h = drop(gelu(a2 W1^T + b1))
            dc = torch.empty_like(s["a2"])
            du = None
            ln2_done = False
            ln2_in_mlp = False
            if fused_mlp:
                # fused branch: h and h' recomputed from a2, all four parameter gradients from one pass -- and norm2's backward on the
                # kernel's fp32 dL/da2: g += dLN, the attention branch's masked copy (written over gm: a tile's rows are read before
                # they are written) and dgamma / dbeta come out of the same launch; no dL/da2 tensor, no LayerNorm-backward launch
                ln2_in_mlp = True
                ops.mlp_bwd(s["d_mlp"], gm, s["a2"], ar.operand(f"{pb}.mlp.fc1.weight"), ar.master(f"{pb}.mlp.fc1.bias"),
                            ar.operand(f"{pb}.mlp.fc2.weight"), None, ar.g(f"{pb}.mlp.fc1.weight"), ar.g(f"{pb}.mlp.fc1.bias"),
                            ar.g(f"{pb}.mlp.fc2.weight"), ar.g(f"{pb}.mlp.fc2.bias"), mask_bits=s["mlp_bits"], partials=self._mlp_partials(s["d_mlp"], dev),
                            ln=dict(x=s["x_mid"], stats=s["st2"], gamma=ar.master(f"{pb}.norm2.weight"), g=g, gm_next=gm,
                                    next_mask=s["d_proj"].out_drop, dgamma=ar.g(f"{pb}.norm2.weight"), dbeta=ar.g(f"{pb}.norm2.bias")))
            else:
                d_fc2_b = ops.linear_desc(cc, M, Cc, 4 * Cc, cc, cc, ACT_GELU)  # dy = gm: operand dtype, already masked
                weight_grad(d_fc2_b, gm, s["h"], ar.g(f"{pb}.mlp.fc2.weight"), ar.g(f"{pb}.mlp.fc2.bias"))
                du = torch.empty_like(s["h"])
                want_ln2 = fuse_ln_bwd and Cc <= ln_bwd_max_c and ops.bwd_data_ln_supported(cc, 4 * Cc, Cc)
                if ops.mlp_wide_bwd_supported(ct, Cc, 4 * Cc):
                    # stages 1-2 (round 6): du = (gm W2) x h' and dL/da2 = du W1 as ONE launch -- du is written once for fc1's weight gradient
                    # and feeds the second product from registers; at 128 channels norm2's backward is the launch's epilogue, as it is
                    # focal_linear_bwd_data_ln's
                    d_wide = ops.mlp_desc(cc, M, Cc, 4 * Cc)
                    if want_ln2:
                        gm_attn = torch.empty_like(gm) if grouped else gm
                        ops.mlp_wide_bwd_data(d_wide, gm, s["hg"], ar.operand(f"{pb}.mlp.fc1.weight"), ar.operand(f"{pb}.mlp.fc2.weight"), du,
                                              ln=dict(x=s["x_mid"], stats=s["st2"], gamma=ar.master(f"{pb}.norm2.weight"), g=g, g_masked=gm_attn,
                                                      mask=s["d_proj"].out_drop, dgamma=ar.g(f"{pb}.norm2.weight"), dbeta=ar.g(f"{pb}.norm2.bias")))
                        ln2_in_mlp = True
                    else:
                        ops.mlp_wide_bwd_data(d_wide, gm, s["hg"], ar.operand(f"{pb}.mlp.fc1.weight"), ar.operand(f"{pb}.mlp.fc2.weight"), du, dc=dc)
                    weight_grad(s["d_fc1"], du, s["a2"], ar.g(f"{pb}.mlp.fc1.weight"), ar.g(f"{pb}.mlp.fc1.bias"))
                else:
                    ops.linear_bwd_data(d_fc2_b, gm, ar.operand(f"{pb}.mlp.fc2.weight"), s["hg"], du)
                    weight_grad(s["d_fc1"], du, s["a2"], ar.g(f"{pb}.mlp.fc1.weight"), ar.g(f"{pb}.mlp.fc1.bias"))
                    if want_ln2:
                        ln2_done = True  # dX of fc1 and norm2's backward in one kernel (below)
                    else:
                        ops.linear_bwd_data(s["d_fc1"], du, ar.operand(f"{pb}.mlp.fc1.weight"), None, dc)
            # (the MLP branch's gm is still an operand of a pending weight gradient unless the fused branch has consumed it)
            if not (ln2_in_mlp and not fused_mlp):
                gm_attn = torch.empty_like(gm) if (not fused_mlp and grouped) else gm
            if ln2_in_mlp:
                pass
            elif ln2_done:
                ops.linear_bwd_data_ln(s["d_fc1"], du, ar.operand(f"{pb}.mlp.fc1.weight"), s["x_mid"], s["st2"], ar.master(f"{pb}.norm2.weight"), g,
                                       ar.g(f"{pb}.norm2.weight"), ar.g(f"{pb}.norm2.bias"), g_masked=gm_attn, mask=s["d_proj"].out_drop)
            else:
                ops.layernorm_bwd(dc, s["x_mid"], s["st2"], ar.master(f"{pb}.norm2.weight"), g, True,
                                  ar.g(f"{pb}.norm2.weight"), ar.g(f"{pb}.norm2.bias"), dx_masked=gm_attn, mask=s["d_proj"].out_drop)
            # ---- attention branch: x_mid = x + mask * (o Wp^T + bp)
            d_proj_b = ops.linear_desc(cc, M, Cc, Cc, cc, cc)
            weight_grad(d_proj_b, gm_attn, s["o"], ar.g(f"{pb}.attn.proj.weight"), ar.g(f"{pb}.attn.proj.bias"))
            do = dc  # reuse the [M, C] CT buffer
            dqkv = torch.empty(M, 3 * Cc, dtype=ct, device=dev)
            # (the branch's backward as ONE launch was built and measured in round 4 and is no faster inside the step: profiles/r4_attn_branch_bwd.txt,
            # tools/lab_attn_branch_bwd/)
            if s["qkv"] is None:
                # 64-channel blocks: q / k / v are recomputed from a1 inside the kernel (see forward), and so is the proj layer's input
                # gradient -- the kernel forms its head's slice of gm_attn . Wproj per item: no dX launch, no dO tensor
                ops.window_attn_qkv_bwd(s["d_att"], s["a1"], ar.operand(f"{pb}.attn.qkv.weight"), ar.master(f"{pb}.attn.qkv.bias"),
                                        ar.master(f"{pb}.attn.relative_position_bias_table"), gm_attn, dqkv,
                                        ar.g(f"{pb}.attn.relative_position_bias_table"), wproj=ar.operand(f"{pb}.attn.proj.weight"))
            else:
                ops.linear_bwd_data(d_proj_b, gm_attn, ar.operand(f"{pb}.attn.proj.weight"), None, do)
                ops.window_attn_bwd(s["d_att"], s["qkv"], ar.master(f"{pb}.attn.relative_position_bias_table"), do, dqkv,
                                    ar.g(f"{pb}.attn.relative_position_bias_table"))
            # the next consumer of g: block k-1's MLP branch, unless a PatchMerging (handled above) or the embedding comes first
            nxt = blocks[k - 1]["d_fc2"].out_drop if (k > 0 and k not in merges) else None
            weight_grad(s["d_qkv"], dqkv, s["a1"], ar.g(f"{pb}.attn.qkv.weight"), ar.g(f"{pb}.attn.qkv.bias"))
            da = do
            ln1_fused = fuse_ln_bwd and Cc <= ln_bwd_max_c and ops.bwd_data_ln_supported(cc, 3 * Cc, Cc)
            if not ln1_fused:
                ops.linear_bwd_data(s["d_qkv"], dqkv, ar.operand(f"{pb}.attn.qkv.weight"), None, da)
            # First block of a stage behind a PatchMerging: the merge's reduction linear takes this block's finished g as its dy.  With
            # a plain `ct` copy of g written by the LayerNorm backward below (as every other linear gets its dy), the reduction's weight
            # gradient joins this block's group launch and its dX reads 2-byte operands.
            mg_dw = merges.get(k) if grouped128 else None
            if mg_dw is not None and ops.dw_group_kind(cc, mg_dw["d_red"].M, mg_dw["d_red"].N, mg_dw["d_red"].K) != 2:
                mg_dw = None
            waiting = False
            if mg_dw is None and dw_items:
                waiting = submit_dw(dw_items, grouped128)
            du = None
            want_gm = nxt is not None or mg_dw is not None
            if want_gm and (mg_dw is not None or waiting):
                gm = torch.empty_like(gm)  # the old buffer is an operand of a group launch that has not run yet
            if ln1_fused:  # dX of qkv and norm1's backward in one kernel
                # the encoder's first block behind the frozen patch embedding: its input is a leaf nobody differentiates, so the
                # residual-stream gradient stops here -- only norm1's dgamma / dbeta are produced (no read / update / re-cast of g)
                g_out = None if (k == 0 and saved.get("embed") is None and not want_gm) else g
                ops.linear_bwd_data_ln(s["d_qkv"], dqkv, ar.operand(f"{pb}.attn.qkv.weight"), s["x"], s["st1"], ar.master(f"{pb}.norm1.weight"), g_out,
                                       ar.g(f"{pb}.norm1.weight"), ar.g(f"{pb}.norm1.bias"), g_masked=gm if want_gm else None, mask=nxt)
            else:
                ops.layernorm_bwd(da, s["x"], s["st1"], ar.master(f"{pb}.norm1.weight"), g, True,
                                  ar.g(f"{pb}.norm1.weight"), ar.g(f"{pb}.norm1.bias"),
                                  dx_masked=gm if want_gm else None, mask=nxt)
            if mg_dw is not None:
                dw_items.append((gm.view(mg_dw["d_red"].M, mg_dw["d_red"].N), mg_dw["a4"], ar.g(f"{mg_dw['pm']}.reduction.weight"), None))
                submit_dw(dw_items, True)
                mg_dw["dw_done"] = True
            del dqkv
            blocks[k] = None  # free this block's activations as we go
            dw_items = dw_calls = weight_grad = None
        flush_dw()
        state.update(g=g, gm=gm, next=stop - 1)

    def _backward_tail(self, saved, state):
        bb, ar = self.bb, self.bb.arena()
        ct = bb.compute_dtype
        g = state["g"]
        # g now holds dL/d(patch-embed tokens).  FOCAL pretraining: the embedding is frozen and its input is a leaf -> stop here.
        es = saved.get("embed")
        if es is not None:  # supervised training: LayerNorm backward, then the convolution's weight / bias gradient
            pe = f"patch_embed.{self.loc}.{self.mod}"
            dz = torch.empty_like(es["z"])
            ops.layernorm_bwd(g.view(es["z"].shape), es["z"], es["st"], ar.master(f"{pe}.norm.weight"), dz, False,
                              ar.g(f"{pe}.norm.weight"), ar.g(f"{pe}.norm.bias"))
            ops.conv_in_bwd_weight(es["d"], es["xpad"], dz, ar.g(f"{pe}.proj.weight"), ar.g(f"{pe}.proj.bias"))


class ProjectorHead:
    """mod_projectors[mod] = Linear -> ReLU -> Linear (models/SW_Transformer.py:157-161, models/DeepSense.py:84-91);
    fp32 activations, `compute` operands."""

    def __init__(self, backbone, mod):
        self.bb, self.mod = backbone, mod
        self.pre = f"mod_projectors.{mod}"

    def forward(self, feat):
        bb, ar = self.bb, self.bb.arena()
        f32 = ops.code(torch.float32)
        cc = f32 if tail_fp32(bb, "projector") else ops.code(bb.compute_dtype)   # (fp32 operands in bf16 mode too: tail_fp32)
        B, K = feat.shape
        E = ar.index[f"{self.pre}.0.weight"][2][0]
        from ._lib import ACT_RELU_OUT, EPI_RELU
        d0 = ops.linear_desc(cc, B, E, K, f32, f32, ACT_NONE, EPI_RELU)
        h = torch.empty(B, E, dtype=torch.float32, device=feat.device)
        ops.linear_fwd(d0, feat, tail_weight(bb, ar, f"{self.pre}.0.weight", "projector"), ar.master(f"{self.pre}.0.bias"), None, h)
        d2 = ops.linear_desc(cc, B, E, E, f32, f32, ACT_RELU_OUT, EPI_NONE)
        z = torch.empty(B, E, dtype=torch.float32, device=feat.device)
        ops.linear_fwd(d2, h, tail_weight(bb, ar, f"{self.pre}.2.weight", "projector"), ar.master(f"{self.pre}.2.bias"), None, z)
        return z, dict(feat=feat, h=h, d0=d0, d2=d2)

    def backward(self, saved, dz):
        ar = self.bb.arena()
        if dz.dtype != torch.float32 or not dz.is_contiguous():
            dz = dz.float().contiguous()
        ops.linear_bwd_weight(saved["d2"], dz, saved["h"], ar.g(f"{self.pre}.2.weight"), ar.g(f"{self.pre}.2.bias"))
        dh = torch.empty_like(saved["h"])
        ops.linear_bwd_data(saved["d2"], dz, tail_weight(self.bb, ar, f"{self.pre}.2.weight", "projector"), saved["h"], dh)  # masked by h > 0
        ops.linear_bwd_weight(saved["d0"], dh, saved["feat"], ar.g(f"{self.pre}.0.weight"), ar.g(f"{self.pre}.0.bias"))
        dfeat = torch.empty_like(saved["feat"])
        ops.linear_bwd_data(saved["d0"], dh, tail_weight(self.bb, ar, f"{self.pre}.0.weight", "projector"), None, dfeat)
        return dfeat
