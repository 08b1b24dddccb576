"""Execution of one (location, modality) DeepSense encoder on the HIP kernels, forward and backward
(reference: models/DeepSense.py:108-157 -> ConvBlock (ConvModules.py:187-216) -> RecurrentBlock (RecurrentModule.py:14-31)).

Layout: activations are channel-last tokens [B*I*S, C] (row = (sample, interval, spectrum bin)), so
  * every [1,k] conv is an MFMA GEMM over a sliding row window, BatchNorm statistics are column sums,
  * the reference's permute + reshape to [B, C*S, I] before the 1x1 Conv1d is a plain row-major view [B*I, S*C]
    (the 1x1 weight is re-ordered from (c*S + s) to (s*C + c) on the fly), and its output [B*I, 128] is already
    the batch-first GRU input.
GRU: input projections for all 10 steps in one GEMM per direction; per step one [B,H]x[H,3H] GEMM + one gate
kernel; backward mirrors it and ends with ONE weight-gradient GEMM per matrix over all steps.
"""
import os

import torch

from . import ops
from ._lib import ACT_NONE, EPI_NONE

# The passes of a step run their backward side by side: their stand-alone weight gradients (the convolutions', the 1x1 output conv's) ask for
# fewer workgroups than the default plan, i.e. fewer fp32 atomics onto their small outputs (include/focal_hip.h focal_linear_desc.dw_workgroups,
# gemm.hpp: focal_dw_plan).  Four passes (round 4): 192 was the optimum (+4 % over the default 512); two passes of both views (round 5), same
# box: 128: 129.9 k, 192: 133.5 k, 256: 133.7 k, 384: 135.5 k, 512: 134.8 k windows/s.  A field of the descriptors, not a process-wide switch.
DW_WORKGROUPS = int(os.environ.get("FOCAL_DS_DW_WORKGROUPS", "384"))
# the GRU's eight weight gradients of a pass as one launch: its workgroup target over all of them (two passes run side by side)
GRU_DW_WORKGROUPS = int(os.environ.get("FOCAL_GRU_DW_WORKGROUPS", "1536"))


class DeepSenseModEncoder:
    def __init__(self, backbone, loc, mod, mod_index):
        self.bb, self.loc, self.mod, self.mod_index = backbone, loc, mod, mod_index
        self.pre = f"loc_mod_extractors.{loc}.{mod}"
        self.rnn = f"recurrent_layers.{mod}.gru"
        self.geo = backbone.geometry[loc][mod]

    def _stream(self, view, uid):
        return ((view * 8 + self.mod_index) * 64 + uid) * 8

    def _sink(self, order, device, groups=1):
        """[BatchNorm layer, {mean, unbiased variance}, C]: where pass `order` of a step records its batch statistics
        ([layer, {mean, variance}, view, C] for a pass that carries both views: order "both")."""
        sinks = self.__dict__.setdefault("_sinks", {})
        key = (order, torch.device(device))
        if key not in sinks:
            shape = (1 + self.geo["n_inter"], 2, self.geo["C"]) if groups == 1 else (1 + self.geo["n_inter"], 2, groups, self.geo["C"])
            sinks[key] = torch.zeros(shape, dtype=torch.float32, device=device)
        return sinks[key]

    def prepare_packs(self):
        """Every re-ordered weight this encoder's passes need -- the [1,k] filters for the forward GEMM and (flipped) for the data
        gradient, the 1x1 output conv, the GRU's W_hh for the backward recurrence -- in ONE launch on the caller's stream, before the
        passes fork: both views of a step read them (13 launches per pass in round 2, ~5 us each on every pass's chain)."""
        bb, ar, geo = self.bb, self.bb.arena(), self.geo
        ct = bb.compute_dtype
        C, S, k, H = geo["C"], geo["S"], geo["k"], geo["H"]
        dev = ar.device
        store = self.__dict__.setdefault("_pack_store", {})

        def dst(key, shape):
            if key not in store or store[key].dtype != ct or store[key].device != dev:
                store[key] = torch.empty(shape, dtype=ct, device=dev)
            return store[key]
        entries = []
        for li in range(geo["n_inter"]):
            w = ar.master(f"{self.pre}.conv_layers_inter.{li}.conv.weight")  # [C, C, 1, k]
            entries.append((w, dst(("fwd", li), (C, k, C)), C, C, k, ops.PACK_PERMUTE))
            entries.append((w, dst(("bwd", li), (C, k, C)), C, C, k, ops.PACK_CONV_BWD))
        n_out = geo["C_out"]
        entries.append((ar.master(f"{self.pre}.conv_layer_out.weight"), dst(("out",), (n_out, S, C)), n_out, C, S, ops.PACK_PERMUTE))
        if ct == torch.bfloat16 and H in (128, 256):
            # W_hh for the sequence kernels in MFMA-fragment order (round 5: their prologue then loads one contiguous KB per instruction
            # instead of sixteen 64-byte row segments): [3H, H] for the forward recurrence, its transpose [H, 3H] for the backward one
            rowmajor = os.environ.get("FOCAL_GRU_WHH_ROWMAJOR") == "1"  # (same-box A/B: the round-4 operands)
            for layer in range(geo["n_rnn"]):
                for suf in ("", "_reverse"):
                    w = ar.master(f"{self.rnn}.weight_hh_l{layer}{suf}")
                    if rowmajor:
                        entries.append((w, dst(("whh", layer, suf), (1, H, 3 * H)), 1, 3 * H, H, ops.PACK_PERMUTE))
                        continue
                    entries.append((w, dst(("whh_frag", layer, suf), (3 * H * H,)), 3 * H, H, 1, ops.PACK_FRAG))
                    entries.append((w, dst(("whh_t_frag", layer, suf), (3 * H * H,)), 3 * H, H, 1, ops.PACK_FRAG_T))
        for lo in range(0, len(entries), ops.PACK_MAX):
            ops.pack_multi(entries[lo:lo + ops.PACK_MAX], ct)
        self._packs = store

    def _packed(self, key, make):
        """The re-ordered weight `key` from this step's prepare_packs, or made on the spot (a caller that runs a pass on its own)."""
        packs = getattr(self, "_packs", None)
        return packs[key] if packs is not None and key in packs else make()

    def _combine_running(self, v1_of, v2_of):
        buf = self.bb.buffer
        names = [f"{self.pre}.conv_layer_in"] + [f"{self.pre}.conv_layers_inter.{li}" for li in range(self.geo["n_inter"])]
        run, v1, v2 = [], [], []
        for i, p in enumerate(names):
            for j, which in enumerate(("running_mean", "running_var")):
                run.append(buf(f"{p}.batch_norm.{which}"))
                v1.append(v1_of(i, j))
                v2.append(v2_of(i, j))
        ops.bn_running_combine(run, v1, v2, 0.1)

    def finish_views(self, device):
        """Both passes of a step have recorded their statistics: apply the two running-buffer updates (view 1's, then view 2's)."""
        if getattr(self, "_sinks_filled", 0) != 3:
            if getattr(self, "_sinks_filled", 0):
                raise ops._lib.FocalHipError("DeepSense: one view's pass recorded BatchNorm statistics and the other did not")
            return
        self._sinks_filled = 0
        s0, s1 = self._sink(0, device), self._sink(1, device)
        self._combine_running(lambda i, j: s0[i, j], lambda i, j: s1[i, j])

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x_freq, view, training):
        bb, ar, geo = self.bb, self.bb.arena(), self.geo
        ct = bb.compute_dtype
        cc, f32 = ops.code(ct), ops.code(torch.float32)
        rng = bb.rng_state() if training else None
        p_drop = bb.drop_rate if training else 0.0
        if x_freq.dtype != torch.float32 or not x_freq.is_contiguous():
            x_freq = x_freq.float().contiguous()
        B, cin, I, S_in = x_freq.shape
        S, C = geo["S"], geo["C"]
        rows = B * I * S
        buf = bb.buffer
        sv = dict(B=B, view=view, x=x_freq, layers=[])
        # The two views of a step run as two passes on their own streams (pass_order 0 / 1).  The reference's BatchNorm running buffers
        # see view 1's statistics, then view 2's; rather than ordering the passes at their last BatchNorm (round 2: view 2's pass
        # started ~0.45 ms late, all of it on the step's critical path), each pass records its batch statistics in a sink of its own
        # (momentum 1: the "running" buffer it is given simply receives the statistic) and DeepSense.finish_views applies both updates
        # afterwards in one launch (ops.bn_running_combine) -- the same two updates, in the reference's order.
        order = getattr(self, "pass_order", None)
        # Round 5: BOTH views in this pass (the batch is view 1's B windows, then view 2's; DeepSense.views_share_pass).  Every BatchNorm
        # keeps one set of batch statistics per view (focal_bn_desc.groups = 2: the reference normalises each backbone call by itself,
        # ConvModules.py:86) and records them in a sink; the two running-buffer updates are applied, in the reference's order, at the end of
        # the convolution stack.  Everything else -- convolutions, the GRU, the projector -- is per window: the same kernels on 2B windows,
        # half the launches of two passes.
        G = int(getattr(self, "views_in_batch", 1) or 1) if training else 1
        if G > 1 and (order is not None or B % G):
            raise ops._lib.FocalHipError(f"DeepSense: a pass with {G} views needs a batch of {G} equal parts and no pass order (B = {B})")
        side_by_side = (order is not None or G > 1) and training
        sink = self._sink(order if G == 1 else "both", x_freq.device, G) if side_by_side else None
        if order == 1 and not side_by_side and getattr(self, "_bn_done", None) is not None:
            torch.cuda.current_stream(x_freq.device).wait_event(self._bn_done)
            self._bn_done = None
        momentum = 1.0 if side_by_side else 0.1

        def running(prefix, i):
            if side_by_side:
                return sink[i, 0], sink[i, 1]
            return buf(f"{prefix}.batch_norm.running_mean"), buf(f"{prefix}.batch_norm.running_var")
        # ---- conv stack
        pin = f"{self.pre}.conv_layer_in"
        d_in = ops.conv_in_desc(B, cin, I, S_in, S, geo["k_in"], geo["stride"], geo["pad_in"], C)
        z = ops.conv_in_fwd(d_in, x_freq, ar.master(f"{pin}.conv.weight"), ar.master(f"{pin}.conv.bias"))
        d_bn = ops.bn_desc(cc, rows, C, I * S, p_drop, rng, self._stream(view, 0), momentum=momentum, groups=G)
        mr = ops.bn_stats(d_bn, z, *running(pin, 0), training, bb.sync_bn and G == 1)
        y, ya = ops.bn_act_fwd(d_bn, z, mr, ar.master(f"{pin}.batch_norm.weight"), ar.master(f"{pin}.batch_norm.bias"), None, ct)
        sv["in"] = dict(d=d_in, z=z, mr=mr, d_bn=d_bn, p=pin)
        if training:
            bb.bump_bn_counters(self.pre, G)  # every BatchNorm of this encoder: num_batches_tracked += 1 per view, one launch
        k = geo["k"]
        d_cv = ops.conv_desc(cc, rows, S, C, C, k, dw_workgroups=DW_WORKGROUPS)
        for li in range(geo["n_inter"]):
            pl = f"{self.pre}.conv_layers_inter.{li}"
            w = ar.master(f"{pl}.conv.weight")  # [C, C, 1, k]
            w_fwd = self._packed(("fwd", li), lambda: ops.permute_pack(w, C, C, k, ct))
            d_bn = ops.bn_desc(cc, rows, C, I * S, p_drop, rng, self._stream(view, 1 + li), momentum=momentum, groups=G)
            # (per-view statistics from the epilogue: a 64- or 128-row tile must lie inside one view)
            if training and ct == torch.bfloat16 and not (bb.sync_bn and ops._sync_world() > 1) and (G == 1 or (rows // G) % 128 == 0):
                # the statistics come out of the convolution's epilogue: z is not read back for them, one launch less per layer
                if ops.conv_fwd_bn_sums_supported(d_cv, d_bn, ya, w_fwd):
                    # ... and are finished by the BatchNorm launch itself: the convolution ends on fire-and-forget adds (round 6)
                    z, sums = ops.conv_fwd_bn_sums(d_cv, ya, w_fwd, ar.master(f"{pl}.conv.bias"), d_bn)
                    y_next, ya_next, mr = ops.bn_act_fwd_sums(d_bn, z, sums, *running(pl, 1 + li), ar.master(f"{pl}.batch_norm.weight"),
                                                              ar.master(f"{pl}.batch_norm.bias"), y, ct)
                    sv["layers"].append(dict(p=pl, z=z, mr=mr, d_bn=d_bn, xa=ya))
                    y, ya = y_next, ya_next
                    continue
                z, mr = ops.conv_fwd_bn(d_cv, ya, w_fwd, ar.master(f"{pl}.conv.bias"), d_bn, *running(pl, 1 + li))
            else:
                z = ops.conv_fwd(d_cv, ya, w_fwd, ar.master(f"{pl}.conv.bias"))
                mr = ops.bn_stats(d_bn, z, *running(pl, 1 + li), training, bb.sync_bn and G == 1)
            y_next, ya_next = ops.bn_act_fwd(d_bn, z, mr, ar.master(f"{pl}.batch_norm.weight"), ar.master(f"{pl}.batch_norm.bias"), y, ct)
            sv["layers"].append(dict(p=pl, z=z, mr=mr, d_bn=d_bn, xa=ya))
            y, ya = y_next, ya_next
        sv["d_cv"] = d_cv
        if G > 1:
            self._combine_running(lambda i, j: sink[i, j, 0], lambda i, j: sink[i, j, 1])
        elif side_by_side:
            self._sinks_filled = getattr(self, "_sinks_filled", 0) | (1 << order)
        elif order == 0:
            self._bn_done = torch.cuda.Event()
            self._bn_done.record(torch.cuda.current_stream(x_freq.device))
        pout = f"{self.pre}.conv_layer_out"
        n_out = geo["C_out"]
        w_out = self._packed(("out",), lambda: ops.permute_pack(ar.master(f"{pout}.weight"), n_out, C, S, ct))  # [n][c*S + s] -> [n][s*C + c]
        d_out = ops.linear_desc(cc, B * I, n_out, S * C, cc, f32, dw_workgroups=DW_WORKGROUPS)
        c_out = torch.empty(B * I, n_out, dtype=torch.float32, device=y.device)
        ops.linear_fwd(d_out, ya, w_out, ar.master(f"{pout}.bias"), None, c_out)
        sv.update(ya_last=ya, w_out=w_out, d_out=d_out, pout=pout)
        # ---- bidirectional GRU
        T, H = I, geo["H"]
        gd = ops.GRUDesc(B, T, H)
        x_l = c_out
        sv["gru"] = []
        for layer in range(geo["n_rnn"]):
            F = x_l.shape[1]
            out = torch.empty(B, T, 2 * H, dtype=torch.float32, device=y.device)
            lsv = dict(x=x_l, dirs=[])
            # bf16 operands: the whole 10-step recurrence of this layer, both directions, is ONE launch (focal_gru_seq_fwd);
            # the exact-fp32 mode keeps one GEMM + one gate kernel per step
            seq = ct == torch.bfloat16 and H in (128, 256)
            for di, suf in enumerate(("", "_reverse")):
                wih, whh = f"{self.rnn}.weight_ih_l{layer}{suf}", f"{self.rnn}.weight_hh_l{layer}{suf}"
                bih, bhh = f"{self.rnn}.bias_ih_l{layer}{suf}", f"{self.rnn}.bias_hh_l{layer}{suf}"
                d_ih = ops.linear_desc(cc, B * T, 3 * H, F, f32, f32, dw_workgroups=DW_WORKGROUPS)
                gi = torch.empty(B * T, 3 * H, dtype=torch.float32, device=y.device)
                ops.linear_fwd(d_ih, x_l, ar.operand(wih), ar.master(bih), None, gi)
                d_hh = ops.linear_desc(cc, B, 3 * H, H, f32, f32)
                hs = torch.zeros(T + 1, B, H, dtype=torch.float32, device=y.device)  # hs[0] = h0 = 0
                save = torch.empty(T, 4, B, H, dtype=torch.float32, device=y.device)
                if not seq:
                    gh = torch.empty(B, 3 * H, dtype=torch.float32, device=y.device)
                    for s in range(T):
                        t = s if di == 0 else T - 1 - s
                        ops.linear_fwd(d_hh, hs[s], ar.operand(whh), ar.master(bhh), None, gh)
                        ops.gru_gate_fwd(gd, t, di * H, gi, gh, hs[s], hs[s + 1], out, save[s])
                lsv["dirs"].append(dict(names=(wih, whh, bih, bhh), d_ih=d_ih, d_hh=d_hh, hs=hs, save=save, gi=gi))
            if seq:
                dirs = lsv["dirs"]
                packs = getattr(self, "_packs", None) or {}
                frag = all(("whh_frag", layer, suf) in packs for suf in ("", "_reverse"))
                whh = [packs[("whh_frag", layer, suf)] for suf in ("", "_reverse")] if frag else [ar.operand(d["names"][1]) for d in dirs]
                ops.gru_seq_fwd(ops.GRUDesc(B, T, H, 1 if frag else 0), [d["gi"] for d in dirs], whh,
                                [ar.master(d["names"][3]) for d in dirs], [d["hs"] for d in dirs], [d["save"] for d in dirs], out)
            for d in lsv["dirs"]:
                del d["gi"]
            lsv["seq"] = seq
            sv["gru"].append(lsv)
            if layer + 1 < geo["n_rnn"]:
                if p_drop > 0:  # nn.GRU applies dropout to the outputs of every layer but the last
                    sid = self._stream(view, 16 + layer)
                    x_l = ops.dropout(out.view(B * T, 2 * H), rng, sid, p_drop)
                    lsv["drop"] = (rng, sid, p_drop)
                else:
                    x_l = out.view(B * T, 2 * H)
        feat = ops.mean_time(out, B, T, 2 * H)
        sv.update(gd=gd, T=T, H=H)
        return feat, sv

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, sv, dfeat):
        bb, ar, geo = self.bb, self.bb.arena(), self.geo
        ct = bb.compute_dtype
        cc, f32 = ops.code(ct), ops.code(torch.float32)
        dev = dfeat.device
        if dfeat.dtype != torch.float32 or not dfeat.is_contiguous():
            dfeat = dfeat.float().contiguous()
        B, T, H, gd = sv["B"], sv["T"], sv["H"], sv["gd"]
        # ---- GRU, last layer first
        dout, ld_b, ld_t, scale = dfeat, 2 * H, 0, 1.0 / T  # d(mean over time): every step gets dfeat / T
        # The GRU's weight gradients (W_hh, W_ih of every layer and direction: fp32 operands, 5 120 rows each) do not feed the backward chain:
        # they are collected and leave as ONE launch behind a problem table (round 5: eight 16 us launches per pass before)
        gru_dw = []
        for layer in range(geo["n_rnn"] - 1, -1, -1):
            lsv = sv["gru"][layer]
            x_l = lsv["x"]
            F = x_l.shape[1]
            dx = None
            bufs = []
            for di, dsv in enumerate(lsv["dirs"]):
                dgi = torch.empty(B * T, 3 * H, dtype=torch.float32, device=dev)
                dgh = torch.empty(T, B, 3 * H, dtype=torch.float32, device=dev)
                bufs.append((dgi, dgh))
            if lsv["seq"]:
                dirs = lsv["dirs"]
                packs = getattr(self, "_packs", None) or {}
                frag = all(("whh_t_frag", layer, suf) in packs for suf in ("", "_reverse"))
                if frag:  # W_hh^T in fragment order (prepare_packs)
                    whh_t = [packs[("whh_t_frag", layer, suf)] for suf in ("", "_reverse")]
                else:     # row-major [H][3H] bf16: FOCAL_GRU_WHH_ROWMAJOR's packs, or made on the spot (a caller that runs a pass on its own)
                    whh_t = [self._packed(("whh", layer, suf), lambda d=d: ops.permute_pack(ar.master(d["names"][1]), 1, 3 * H, H, ct))
                             for d, suf in zip(dirs, ("", "_reverse"))]
                ops.gru_seq_bwd(ops.GRUDesc(B, T, H, 1 if frag else 0), dout, ld_b, ld_t, scale, whh_t, [d["hs"] for d in dirs],
                                [d["save"] for d in dirs], [b_[0] for b_ in bufs], [b_[1] for b_ in bufs])
            for di, dsv in enumerate(lsv["dirs"]):
                wih, whh, bih, bhh = dsv["names"]
                hs, save = dsv["hs"], dsv["save"]
                dgi, dgh = bufs[di]
                if not lsv["seq"]:
                    dhz = torch.empty(2, B, H, dtype=torch.float32, device=dev)
                    dh_rec = torch.empty(B, H, dtype=torch.float32, device=dev)
                    have = False
                    for s in range(T - 1, -1, -1):
                        t = s if di == 0 else T - 1 - s
                        ops.gru_gate_bwd(gd, t, di * H, dout, ld_b, ld_t, scale, dh_rec if have else None, dhz[(s + 1) & 1] if have else None,
                                         save[s], hs[s], dgi, dgh[s], dhz[s & 1])
                        if s > 0:
                            ops.linear_bwd_data(dsv["d_hh"], dgh[s], ar.operand(whh), None, dh_rec)
                            have = True
                gru_dw.append((dgh.view(T * B, 3 * H), hs[:T].reshape(T * B, H), ar.g(whh), ar.g(bhh)))
                gru_dw.append((dgi, x_l, ar.g(wih), ar.g(bih)))
                dxi = torch.empty(B * T, F, dtype=torch.float32, device=dev)
                ops.linear_bwd_data(dsv["d_ih"], dgi, ar.operand(wih), None, dxi)
                if dx is None:
                    dx = dxi
                else:
                    ops.axpy(1.0, dxi, dx)
            if layer > 0:
                prev = sv["gru"][layer - 1]
                if "drop" in prev:
                    rng, sid, p = prev["drop"]
                    dx = ops.dropout(dx, rng, sid, p)
                dout, ld_b, ld_t, scale = dx, T * 2 * H, 2 * H, 1.0
        for lo in range(0, len(gru_dw), ops.DW_TAIL_MAX):
            ops.linear_bwd_weight_group_f32(cc, gru_dw[lo:lo + ops.DW_TAIL_MAX], GRU_DW_WORKGROUPS)
        del gru_dw
        # ---- flatten + 1x1 output conv
        C, S, I = geo["C"], geo["S"], T
        rows = B * I * S
        pout, d_out = sv["pout"], sv["d_out"]
        n_out = geo["C_out"]
        unpack = []  # packed weight gradients, folded back into the arena's layout by ONE launch at the end of the pass
        dwp = ops.zeros((n_out, S * C), dev)
        ops.linear_bwd_weight(d_out, dx, sv["ya_last"], dwp, ar.g(f"{pout}.bias"))
        unpack.append((dwp, ar.g(f"{pout}.weight"), n_out, C, S))
        d_out_data = ops.linear_desc(cc, B * I, n_out, S * C, f32, f32)  # gradient w.r.t. the fp32 residual stream
        g = torch.empty(rows, C, dtype=torch.float32, device=dev)
        ops.linear_bwd_data(d_out_data, dx, sv["w_out"], None, g)
        # ---- residual conv layers
        d_cv, k = sv["d_cv"], geo["k"]
        for li in range(geo["n_inter"] - 1, -1, -1):
            L = sv["layers"][li]
            pl = L["p"]
            dz = ops.bn_act_bwd(L["d_bn"], L["z"], g, L["mr"], ar.master(f"{pl}.batch_norm.weight"), ar.master(f"{pl}.batch_norm.bias"),
                                ar.g(f"{pl}.batch_norm.weight"), ar.g(f"{pl}.batch_norm.bias"), ct, bb.sync_bn)
            dwp = ops.zeros((C, k * C), dev)
            ops.conv_bwd_weight(d_cv, dz, L["xa"], dwp, ar.g(f"{pl}.conv.bias"))
            unpack.append((dwp, ar.g(f"{pl}.conv.weight"), C, C, k))
            w_bwd = self._packed(("bwd", li), lambda: ops.conv_pack_bwd(d_cv, ar.master(f"{pl}.conv.weight"), ct))  # flipped taps, [C_in][k][C_out]
            ops.conv_bwd_data(d_cv, dz, w_bwd, g, g)  # g <- g + conv^T(dz), in place
            sv["layers"][li] = None
        Lin = sv["in"]
        pin = Lin["p"]
        dz = ops.bn_act_bwd(Lin["d_bn"], Lin["z"], g, Lin["mr"], ar.master(f"{pin}.batch_norm.weight"), ar.master(f"{pin}.batch_norm.bias"),
                            ar.g(f"{pin}.batch_norm.weight"), ar.g(f"{pin}.batch_norm.bias"), ct, bb.sync_bn)
        ops.conv_in_bwd_weight(Lin["d"], sv["x"], dz, ar.g(f"{pin}.conv.weight"), ar.g(f"{pin}.conv.bias"))
        ops.unpack_add_multi(unpack)
        # the spectrum is a leaf: nothing flows further
