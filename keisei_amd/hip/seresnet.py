"""HIP execution engine for the SE-ResNet (forward + hand-written backward).

Drives libkeisei_amd.so for CUDA/HIP tensors: activations live in NHWC (B,81,C) as bf16 (AMP on)
or fp32 (AMP off), convolutions run on the MFMA kernels with weights re-packed into fragment
order as *derived caches* (the stored parameters keep the reference's shapes and names), and the
whole network is a single autograd node whose backward is the explicit kernel sequence below --
there is no eager/PyTorch arithmetic on this path.

Kernel sequence per GlobalPoolBiasBlock (reference: se_resnet.py:68-90)
  forward : conv1(+BN1 stats) | bn_coeffs | global_fc (2 gemm) | conv2 with fused
            relu(bn1(.))+g input transform (+BN2 stats, SE squeeze) | bn_coeffs | SE FCs (2 gemm)
            | tail: relu(bn2(.)*sigmoid+shift+x) fused with the next block's global pool
  backward: tail_bwd_reduce | SE FC backward | tail_bwd_dz (+BN2 sums) | bn_bwd_coeffs/apply |
            conv2 dgrad (+ dg sums) | conv2 wgrad (input transform recomputed on the fly) |
            global_fc backward | relu_bn_bwd_reduce | bn_bwd_coeffs/apply | conv1 dgrad |
            conv1 wgrad | block_dx (residual + mean/max/std pool backward)
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os
import threading

import torch
import torch.distributed as dist
from torch import nn

from keisei_amd import _lib

_call = _lib.call


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _Saved:
    """Activations kept between forward and backward."""
    __slots__ = ("B", "T", "train", "xin", "stem", "blocks", "heads", "sync")


# split-K slabs of the row-reduction GEMMs (weight / bias gradients over B or B*81 rows): more, shorter workgroups -- the
# B*81-row policy-head forms are latency-bound chains of K-tiles (measured 64 -> 256: -0.6 % step time)
_FC_SPLITS = int(os.environ.get("KA_FC_SPLITS", "256"))


class SEResNetEngine:
    def __init__(self, model: nn.Module) -> None:
        self.model = model
        self._pack_sets: Dict[tuple, dict] = {}     # (dtype, device, parameter storage) -> pack buffers + job table
        self._scratch: Optional[torch.Tensor] = None
        self._redws: Optional[torch.Tensor] = None
        self._side = None
        self._pack_tkey = None
        self._graphs = {}             # (batch, dtype, parameter storage) -> captured eval forward (rollout inference path)
        self._graph_lock = threading.Lock()
        self._wslab: Optional[torch.Tensor] = None
        self._fc_jobs = None          # open list of deferred FC weight-gradient jobs during a backward pass
        self._row_ring = None
        self._evalc = None            # (key, device table, {id(bn): (scale, shift)}): all eval BatchNorm coefficients, one launch
        self._evalc_sets = {}
        self._fc_tr_sets = {}                       # transposed global_fc weights of the backward chain
        self._tensor_lists = None                   # (buffers, parameters, structure version) of the eval-graph key
        self._tower_tabs = {}                       # pointer tables of the one-launch eval tower (kept: graphs read them)
        self._evalc_live = None
        # Weight gradients on a second stream beside the data-gradient chain: off by default since round 3.  The MFMA kernels are
        # power-bound: two of them sharing the chip finish no sooner than one after the other, and with the
        # main queue's small kernels shortened the second stream no longer fills anything (110.6 / 111.2 ms without it against
        # 111.0 / 111.2 with, one job) -- while every launch now runs at its stand-alone rate (data gradient 0.45 instead of
        # 0.73 ms in the step's trace).  KA_WGRAD_OVERLAP=1 brings the two-stream schedule back.
        self.overlap_wgrad = os.environ.get("KA_WGRAD_OVERLAP", "0") != "0"
        self.fork_fc = os.environ.get("KA_FC_FORK", "1") != "0"      # small-batch forward: global-pool FC chain beside conv1
        self._fc_side = None
        self._gpool_done = None
        self._redcnt = None
        # BatchNorm statistics: stage-1 reduce and coefficient kernel as one launch (KA_BN_ONE_LAUNCH=1; measured 0.8 % slower than
        # the two launches -- the device-scope fences of 256 workgroups cost more than a launch boundary -- so off by default)
        self.bn_one_launch = os.environ.get("KA_BN_ONE_LAUNCH", "0") == "1"
        self._in_forward = False
        self.kernel_events = None       # bench.py: {"conv3x3": [...], "wgrad": [...]} event pairs per launch
        self.weights_epoch = 0          # bumped by the fused optimiser (raw-pointer updates bypass _version)
        self.grad_reducer = None        # OverlappedGradReducer while a fused DDP step runs (hip/grad_reducer.py)
        self.sync_events = None         # bench.py --dist-dry-run: event pairs around every wait for a SyncBN all-reduce

    # ------------------------------------------------------------------ helpers
    def _timed(self, kind: str, name: str, *args) -> None:
        """Launch `name`; when bench.py set ``kernel_events`` bracket the launch with events on the launch stream."""
        ev = self.kernel_events
        if ev is None:
            _call(name, *args)
            return
        stream = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        _call(name, *args)
        b.record(stream)
        ev[kind].append((a, b))
        if kind == "conv3x3" and name == "ka_conv3x3_fwd" and "conv3x3_fwd" in ev and self._in_forward:
            ev["conv3x3_fwd"].append((a, b))

    def notify_weights_updated(self) -> None:
        self.weights_epoch += 1

    def _red_ws(self, C: int, device) -> torch.Tensor:
        n = _lib.query("ka_reduce_workspace_doubles", C)
        if self._redws is None or self._redws.numel() < n or self._redws.device != device:
            self._redws = torch.empty(n, dtype=torch.float64, device=device)
        return self._redws

    def _red_counters(self, device) -> torch.Tensor:
        """arrival counters of the one-launch statistics kernels (zero between launches: the kernels reset them)"""
        if self._redcnt is None or self._redcnt.device != device:
            self._redcnt = torch.zeros(64, dtype=torch.int32, device=device)
        return self._redcnt

    def _wgrad_side(self, n: int, device):
        """(side stream, partial-slab buffer) of the weight-gradient GEMMs.  In backward they are issued on a second
        HIP stream right after the data-gradient conv that shares their input, so the MFMA-bound wgrad overlaps the
        HBM-bound BatchNorm / pooling / FC kernels of the main stream instead of serialising with them."""
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device)
        if self._wslab is None or self._wslab.numel() < n or self._wslab.device != device:
            self._wslab = torch.empty(n, dtype=torch.float32, device=device)
        return self._side, self._wslab

    def _wgrad_launch(self, side, main, inputs, *args) -> None:
        """ka_conv3x3_wgrad on the side stream, ordered after everything already queued on the main stream."""
        if side is None:
            self._timed("wgrad", "ka_conv3x3_wgrad", *args, _lib.stream_ptr(main.device))
            return
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            self._timed("wgrad", "ka_conv3x3_wgrad", *args, _lib.stream_ptr(main.device))
        for t in inputs:                 # main-stream allocations read by the side stream: defer their reuse
            t.record_stream(side)

    def _scratch_f32(self, n: int, device) -> torch.Tensor:
        if self._scratch is None or self._scratch.numel() < n or self._scratch.device != device:
            self._scratch = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        return self._scratch

    def _conv_layers(self):
        m = self.model
        yield "input_conv", m.input_conv
        for i, blk in enumerate(m.blocks):
            yield f"blocks.{i}.conv1", blk.conv1
            yield f"blocks.{i}.conv2", blk.conv2

    def _get_packs(self, T: torch.dtype, device) -> Dict[str, tuple]:
        """fragment-ordered weight copies: name -> (forward pack, dgrad pack | None).

        One set of pack buffers (plus its device-side job table) per (dtype, device, parameter storage), kept for the
        life of the engine: a captured eval graph reads the buffers it was captured with, so alternating dtypes (an fp32
        eval between two bf16 ones) must refresh each set in place and never free or reallocate one."""
        convs = list(self._conv_layers())
        tkey = (T, str(device), tuple(c.weight.data_ptr() for _, c in convs))
        ent = self._pack_sets.get(tkey)
        if ent is None:
            if len(self._pack_sets) >= 4:            # parameters were re-allocated repeatedly: drop stale sets AND the
                self._pack_sets.clear()              # graphs that read them
                self._graphs.clear()
            cpk = 32 if T == torch.bfloat16 else 16
            packs, jobs, mx = {}, [], 0
            for name, conv in convs:
                w = conv.weight
                co, ci = w.shape[0], w.shape[1]
                ci_pad = _round_up(ci, 64) if name == "input_conv" else ci   # only the 50-plane stem input is padded
                nf = 9 * (ci_pad // cpk) * (co // 16) * 64
                fwd = torch.empty(nf * 16, dtype=torch.uint8, device=device)
                jobs.append([w.data_ptr(), fwd.data_ptr(), co, ci, co, ci_pad, 0, 0]); mx = max(mx, nf)
                dg = None
                if name != "input_conv":
                    nd = 9 * (co // cpk) * (ci // 16) * 64
                    dg = torch.empty(nd * 16, dtype=torch.uint8, device=device)
                    jobs.append([w.data_ptr(), dg.data_ptr(), co, ci, ci, co, 1, 0]); mx = max(mx, nd)
                packs[name] = (fwd, dg)
            ent = {"packs": packs, "table": torch.tensor(jobs, dtype=torch.int64).to(device), "max": mx, "key": None}
            self._pack_sets[tkey] = ent
        self._pack_tkey = tkey
        key = (self.weights_epoch, sum(c.weight._version for _, c in convs))
        if key != ent["key"]:
            # after an optimiser step only the single multi-layer pack launch is repeated
            _call("ka_pack_conv3x3_multi", ent["table"], ent["table"].shape[0], ent["max"], _lib.dtype_code(T),
                  _lib.stream_ptr(device))
            ent["key"] = key
        return ent["packs"]

    def _tower_table(self, T, device, packs):
        """Device table of the one-launch eval tower (one row of 14 pointers per block), or None when that kernel does
        not cover the configuration.  Rebuilt when any of the buffers it points at was re-allocated."""
        m = self.model
        if os.environ.get("KA_TOWER", "1") == "0" or self._evalc_live is None or T != torch.bfloat16:
            return None
        blk0 = m.blocks[0]
        C, G, R = m.params.channels, blk0.global_fc[0].out_features, blk0.se_fc1.out_features
        if not _lib.query("ka_tower_eval_supported", C, G, R, _lib.dtype_code(T)):
            return None
        rows = []
        for i, blk in enumerate(m.blocks):
            lins = (blk.global_fc[0], blk.global_fc[2], blk.se_fc1, blk.se_fc2)
            if (id(blk.bn1) not in self._evalc_live or id(blk.bn2) not in self._evalc_live
                    or any(l.bias is None or not l.weight.is_contiguous() or l.weight.dtype != torch.float32 for l in lins)):
                return None
            sc1, sh1 = self._evalc_live[id(blk.bn1)]
            sc2, sh2 = self._evalc_live[id(blk.bn2)]
            rows.append([packs[f"blocks.{i}.conv1"][0].data_ptr(), packs[f"blocks.{i}.conv2"][0].data_ptr(),
                         sc1.data_ptr(), sh1.data_ptr(), sc2.data_ptr(), sh2.data_ptr(),
                         lins[0].weight.data_ptr(), lins[0].bias.data_ptr(), lins[1].weight.data_ptr(), lins[1].bias.data_ptr(),
                         lins[2].weight.data_ptr(), lins[2].bias.data_ptr(), lins[3].weight.data_ptr(), lins[3].bias.data_ptr()])
        key = (str(device), tuple(map(tuple, rows)))
        ent = self._tower_tabs.get(key)
        if ent is None:
            if len(self._tower_tabs) >= 4:               # buffers were re-allocated repeatedly: drop stale tables and
                self._tower_tabs.clear()                 # the graphs that read them
                self._graphs.clear()
            ent = torch.tensor(rows, dtype=torch.int64).to(device)
            self._tower_tabs[key] = ent
        return ent

    def _eval_coeffs_all(self, device, st):
        """Eval-mode scale/shift of EVERY BatchNorm layer from the live running statistics in one launch (the rollout
        forward otherwise spends 82 tiny launches on them).  Returns {id(bn): (scale, shift)} or None when a layer has
        no affine parameters / running statistics (then each layer computes its own)."""
        import struct

        bns = [b for b in self.model.modules() if isinstance(b, nn.modules.batchnorm._BatchNorm)]
        if not bns or any(b.weight is None or b.bias is None or b.running_mean is None or b.running_var is None for b in bns):
            return None
        key = (str(device), tuple((b.weight.data_ptr(), b.bias.data_ptr(), b.running_mean.data_ptr(), b.running_var.data_ptr(),
                                   float(b.eps)) for b in bns))
        ent = self._evalc_sets.get(key)
        if ent is None:
            if len(self._evalc_sets) >= 4:
                self._evalc_sets.clear()
                self._graphs.clear()                 # captured graphs read the coefficient buffers of their set
            mx = max(b.num_features for b in bns)
            out = torch.empty(len(bns), 2, mx, device=device)
            rows, views = [], {}
            for i, b in enumerate(bns):
                C = b.num_features
                sc, sh = out[i, 0, :C], out[i, 1, :C]
                eps_bits = struct.unpack("<I", struct.pack("<f", float(b.eps)))[0]
                rows.append([b.weight.data_ptr(), b.bias.data_ptr(), b.running_mean.data_ptr(), b.running_var.data_ptr(),
                             sc.data_ptr(), sh.data_ptr(), C, eps_bits])
                views[id(b)] = (sc, sh)
            ent = (key, torch.tensor(rows, dtype=torch.int64).to(device), views, mx, out)
            self._evalc_sets[key] = ent
        self._evalc = ent
        _, table, views, mx, _ = self._evalc
        _call("ka_bn_eval_coeffs_multi", table, table.shape[0], mx, st)
        return views

    @staticmethod
    def _sync_group(bn: nn.Module):
        if isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
            # KA_FORCE_COLLECTIVES=1: a world of one rank still issues every collective (the RCCL rehearsal on one GPU)
            return dist.get_world_size() > 1 or os.environ.get("KA_FORCE_COLLECTIVES", "0") == "1"
        return False

    def _allreduce_async(self, t):
        """SyncBN statistics vector: all-reduce issued asynchronously (RCCL: on the process group's own stream, ordered after
        the kernels already queued here), so that the independent launches the caller queues next run under it; the caller
        hands the returned handle to ``_allreduce_wait`` right before the first kernel that reads ``t``."""
        return dist.all_reduce(t, async_op=True)

    def _allreduce_wait(self, work) -> None:
        if work is None:
            return
        ev = self.sync_events
        if ev is None or not torch.cuda.is_available():
            work.wait()
            return
        # bench.py --dist-dry-run: how long the compute stream actually stalls on each statistics collective
        stream = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        work.wait()
        b.record(stream)
        ev.append((a, b))

    def _bn_forward_begin(self, bn, bsum, rows_b, sq, rows_s, C, count, train, device, st):
        """first half of a BatchNorm layer's statistics: the stage-1 reduce and, under SyncBatchNorm, the asynchronous
        all-reduce of the packed vector.  Launches queued between this and ``_bn_forward_end`` overlap the collective."""
        if not train:
            return (bn, None, None, None, C, count, train, device, st)
        sync = self._sync_group(bn)
        ws = self._red_ws(C, device)
        sums = work = None
        if sync:
            # [sum y | sum y^2 | count] in ONE vector = one all-reduce per layer (the count rides along, written by the
            # reduce kernel itself); the dependency chain conv -> statistics -> next conv forbids merging layers
            sums = torch.empty(2 * C + 1, dtype=torch.float64, device=device)
            _call("ka_sync_reduce", bsum, rows_b, sq, rows_s, C, float(count), sums, None, ws, st)
            work = self._allreduce_async(sums)
        elif self.bn_one_launch and C <= 64 * 64:
            return (bn, None, None, (ws, bsum, rows_b, sq, rows_s), C, count, train, device, st)     # one launch, in _bn_forward_end
        else:
            _call("ka_bn_reduce", bsum, rows_b, sq, rows_s, C, None, ws, st)     # stage 1 only: partials stay in ws
        return (bn, sums, work, ws, C, count, train, device, st)

    def _bn_forward_end(self, ctx):
        """returns scale, shift, mean, invstd (fp32 [C])"""
        bn, sums, work, ws, C, count, train, device, st = ctx
        if not train and self._evalc_live is not None and id(bn) in self._evalc_live:
            scale, shift = self._evalc_live[id(bn)]
            return scale, shift, None, None
        scale = torch.empty(C, device=device); shift = torch.empty(C, device=device)
        if not train:
            _call("ka_bn_eval_coeffs", bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), scale, shift, C, st)
            return scale, shift, None, None
        mean = torch.empty(C, device=device); invstd = torch.empty(C, device=device)
        track = bn.track_running_stats and bn.running_mean is not None
        if track and bn.momentum is None:
            momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)      # cumulative average (host sync; non-default)
        else:
            momentum = float(bn.momentum if bn.momentum is not None else 0.0)
        rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if track else (None, None, None)
        if sums is not None:
            self._allreduce_wait(work)
            _call("ka_bn_coeffs", sums, float(count), sums[2 * C:], bn.weight, bn.bias, rm, rv, nbt, momentum, float(bn.eps),
                  scale, shift, mean, invstd, C, st)
        elif isinstance(ws, tuple):
            ws, bsum, rows_b, sq, rows_s = ws
            _call("ka_bn_reduce_coeffs", bsum, rows_b, sq, rows_s, C, ws, self._red_counters(device), float(count), bn.weight, bn.bias,
                  rm, rv, nbt, momentum, float(bn.eps), scale, shift, mean, invstd, st)
        else:
            _call("ka_bn_coeffs_parts", ws, float(count), bn.weight, bn.bias, rm, rv, nbt, momentum, float(bn.eps),
                  scale, shift, mean, invstd, C, st)
        return scale, shift, mean, invstd

    def _bn_forward(self, bn, bsum, rows_b, sq, rows_s, C, count, train, device, st):
        return self._bn_forward_end(self._bn_forward_begin(bn, bsum, rows_b, sq, rows_s, C, count, train, device, st))

    def _gemm(self, A, Bm, C, bias, M, N, K, lda, ldb, ldc, ta, tb, st, abf=0, bbf=0, cbf=0, relu=0, acc=0, ns=1):
        _call("ka_gemm", A, Bm, C, bias, M, N, K, lda, ldb, ldc, ta, tb, abf, bbf, cbf, relu, acc, ns, st)

    def _linear(self, x, lin: nn.Linear, relu: int, st):
        """y = act(x[:, :K] W^T + b); x may be wider than K (pooled statistics carry a 4th, non-feature plane)."""
        M, lda = x.shape
        N, K = lin.weight.shape
        y = torch.empty(M, N, device=x.device)
        self._gemm(x, lin.weight, y, lin.bias, M, N, K, lda, K, N, 0, 1, st, relu=relu)
        return y

    def _fc_chain(self, x, lin1: nn.Linear, lin2: nn.Linear, st, keep: bool, affine=None):
        """(x', hidden, y) of y = lin2(relu(lin1(x'))): one fused launch when the shape allows, else two GEMMs.
        ``affine`` = (scale, shift, alpha): x' = scale * (x * alpha) + shift (returned when kept), otherwise x' = x."""
        M, ldx = x.shape
        H, K1 = lin1.weight.shape
        N2 = lin2.weight.shape[0]
        dev = x.device
        fused = (os.environ.get("KA_FC_CHAIN", "1") != "0" and lin1.weight.is_contiguous() and lin2.weight.is_contiguous()
                 and _lib.query("ka_fc_chain_supported", K1, ldx, H, N2))
        if not fused:
            xp = x
            if affine is not None:
                xp = torch.empty(M, K1, device=dev)
                _call("ka_affine_rows", x, affine[0], affine[1], float(affine[2]), xp, M, K1, st)
            hidden = self._linear(xp, lin1, 1, st)
            return (xp if affine is not None else None), hidden, self._linear(hidden, lin2, 0, st)
        y = torch.empty(M, N2, device=dev)
        hidden = torch.empty(M, H, device=dev) if keep else None
        xp = torch.empty(M, K1, device=dev) if (keep and affine is not None) else None
        sc, sh, alpha = affine if affine is not None else (None, None, 1.0)
        _call("ka_fc_chain", x, sc, sh, float(alpha), lin1.weight, lin1.bias, lin2.weight, lin2.bias, xp, hidden, y,
              M, K1, ldx, H, N2, st)
        return xp, hidden, y

    def _fc_transposed(self, device, st):
        """Transposed copies of every block's global_fc weights for the one-launch backward chain (ka_fc_chain_bwd):
        [(W2^T (G, C), W1^T (3C, G))] per block, refreshed by ONE launch when the weights changed."""
        blocks = self.model.blocks
        ptrs = tuple((b.global_fc[0].weight.data_ptr(), b.global_fc[2].weight.data_ptr()) for b in blocks)
        ent = self._fc_tr_sets.get((str(device), ptrs))
        if ent is None:
            if len(self._fc_tr_sets) >= 4:
                self._fc_tr_sets.clear()
            views, rows, mt = [], [], 1
            for b in blocks:
                w1, w2 = b.global_fc[0].weight, b.global_fc[2].weight            # (G, 3C), (C, G)
                w2t = torch.empty(w2.shape[1], w2.shape[0], device=device); w1t = torch.empty(w1.shape[1], w1.shape[0], device=device)
                views.append((w2t, w1t))
                for src, dst in ((w2, w2t), (w1, w1t)):
                    rows.append([src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]])
                    mt = max(mt, ((src.shape[0] + 31) // 32) * ((src.shape[1] + 31) // 32))
            ent = {"views": views, "table": torch.tensor(rows, dtype=torch.int64).to(device), "n": len(rows), "mt": mt, "key": None}
            self._fc_tr_sets[(str(device), ptrs)] = ent
        key = (self.weights_epoch, sum(b.global_fc[0].weight._version + b.global_fc[2].weight._version for b in blocks))
        if key != ent["key"]:
            _call("ka_transpose_multi", ent["table"], ent["n"], ent["mt"], st)
            ent["key"] = key
        return ent["views"]

    def _gpool_bwd(self, i, blk, dg, g1, bpool, grads, pre, st, tr, side=None):
        """input gradient of the global-pool bias chain (+ its deferred weight / bias gradients): dg (B, C) -> dpool (B, 3C).
        `side` = (stream, event already recorded on the main stream): the chain launch goes to that stream behind the event
        and self._gpool_done is the event to wait for before dpool / dg1 are read (KA_FC_BWD_SIDE=1, experiment)."""
        lin1, lin2 = blk.global_fc[0], blk.global_fc[2]
        B = dg.shape[0]
        if tr is not None:
            w2t, w1t = tr[i]
            dg1 = torch.empty(B, lin1.out_features, device=dg.device)
            dpool_x = torch.empty(B, lin1.in_features, device=dg.device)
            if side is not None:
                sstream, ev = side
                main = torch.cuda.current_stream(dg.device)
                with torch.cuda.stream(sstream):
                    sstream.wait_event(ev)
                    _call("ka_fc_chain_bwd", dg, g1, w2t, w1t, dg1, dpool_x, B, lin2.out_features, lin1.out_features,
                          lin1.in_features, _lib.stream_ptr(dg.device))
                    done = torch.cuda.Event(); done.record(sstream)
                for t in (dg, g1, dg1, dpool_x):
                    t.record_stream(sstream)
                self._gpool_done = done
            else:
                _call("ka_fc_chain_bwd", dg, g1, w2t, w1t, dg1, dpool_x, B, lin2.out_features, lin1.out_features, lin1.in_features, st)
            self._linear_bwd(dg, g1, lin2, grads, pre + "global_fc.2.weight", pre + "global_fc.2.bias", st, need_dx=False)
            self._linear_bwd(dg1, bpool, lin1, grads, pre + "global_fc.0.weight", pre + "global_fc.0.bias", st, need_dx=False)
            return dpool_x
        dg1 = self._linear_bwd(dg, g1, lin2, grads, pre + "global_fc.2.weight", pre + "global_fc.2.bias", st)
        _call("ka_relu_mask", dg1, g1, dg1.numel(), st)
        return self._linear_bwd(dg1, bpool, lin1, grads, pre + "global_fc.0.weight", pre + "global_fc.0.bias", st)

    def _upload_rows(self, rows, device):
        """int64 table -> device through one of two rotating pinned buffers (the host never waits for the stream)."""
        flat = torch.tensor(rows, dtype=torch.int64).reshape(-1)
        ring = self._row_ring
        if ring is None or ring["host"][0].numel() < flat.numel():
            n = max(flat.numel(), 4096)
            ring = {"host": [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(2)],
                    "dev": [torch.empty(n, dtype=torch.int64, device=device) for _ in range(2)], "evt": [None, None], "flip": 0}
            self._row_ring = ring
        i = ring["flip"]
        ring["flip"] ^= 1
        if ring["evt"][i] is not None:
            ring["evt"][i].synchronize()
        ring["host"][i][:flat.numel()].copy_(flat)
        ring["dev"][i][:flat.numel()].copy_(ring["host"][i][:flat.numel()], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["evt"][i] = ev
        return ring["dev"][i]

    def _flush_fc_jobs(self, grads, device, st) -> None:
        """One grouped launch for every deferred FC weight / bias gradient of this backward pass."""
        jobs, self._fc_jobs = self._fc_jobs, None
        if not jobs:
            return None
        total = sum(N * K + (N if bname is not None else 0) for (_, _, _, _, _, N, K, _, bname, _) in jobs)
        flat = torch.empty(total, device=device)
        rows, off, wg = [], 0, 0
        for dy, x, ldx, xbf, M, N, K, wname, bname, wshape in jobs:
            dW = flat[off:off + N * K]; off += N * K
            db_ptr = 0
            if bname is not None:
                db = flat[off:off + N]; off += N
                grads[bname] = db
                db_ptr = db.data_ptr()
            grads[wname] = dW.view(wshape)
            rows.append([dy.data_ptr(), x.data_ptr(), dW.data_ptr(), db_ptr, M, N, K, ldx, xbf, wg])
            wg += ((N + 63) // 64) * ((K + (1 if bname is not None else 0) + 63) // 64)
        table = self._upload_rows(rows, device)
        _call("ka_gemm_grouped_wgrad", table, len(jobs), wg, st)
        return flat

    def _linear_bwd(self, dy, x, lin: nn.Linear, grads, wname, bname, st, need_dx=True, x_bf16=0, dx_out=None, acc_dx=0,
                    defer=True):
        """dW = dy^T x (split over rows), db = colsum(dy), dx = dy W."""
        M, N = dy.shape
        K = lin.weight.shape[1] if lin.weight.ndim == 2 else lin.weight.shape[1]
        dev = dy.device
        ldx = x.shape[1]
        if (defer and self._fc_jobs is not None and M <= 16384 and dy.dtype == torch.float32 and dy.is_contiguous()
                and x.is_contiguous()):    # (the B*81-row policy-head GEMMs keep their split-K launches)
            # weight / bias gradient deferred to the grouped launch at the end of the backward (nothing reads them earlier)
            self._fc_jobs.append((dy, x, ldx, int(x_bf16), M, N, K, wname, bname, tuple(lin.weight.shape)))
            if not need_dx:
                return None
            dx = dx_out if dx_out is not None else torch.empty(M, K, device=dev)
            self._gemm(dy, lin.weight, dx, None, M, K, N, N, K, K, 0, 0, st, acc=acc_dx)
            return dx
        ns = max(1, min(_FC_SPLITS, (M + 511) // 512))
        dW = torch.empty(N, K, device=dev)
        if ns == 1:
            self._gemm(dy, x, dW, None, N, K, M, N, ldx, K, 1, 0, st, bbf=x_bf16)
        else:
            slab = self._scratch_f32(ns * N * K, dev)
            self._gemm(dy, x, slab, None, N, K, M, N, ldx, K, 1, 0, st, bbf=x_bf16, ns=ns)
            _call("ka_reduce_slabs", slab, dW, ns, N * K, 0, st)
        grads[wname] = dW.view_as(lin.weight)
        if bname is not None:
            nsb = max(1, min(_FC_SPLITS, (M + 127) // 128))
            db = torch.empty(N, device=dev)
            if nsb == 1:
                _call("ka_colsum", dy, None, db, None, M, N, 1, st)
            else:
                part = self._scratch_f32(nsb * N, dev)
                _call("ka_colsum", dy, None, part, None, M, N, nsb, st)
                _call("ka_reduce_slabs", part, db, nsb, N, 0, st)
            grads[bname] = db
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else torch.empty(M, K, device=dev)
        self._gemm(dy, lin.weight, dx, None, M, K, N, N, K, K, 0, 0, st, acc=acc_dx)
        return dx

    # ------------------------------------------------------------------ rollout inference: graph-captured eval forward
    def forward_eval_graphed(self, obs: torch.Tensor, T: torch.dtype):
        """Eval-mode forward without saved activations, replayed from a captured HIP graph.  The small-batch rollout
        forward (select_actions: N = number of environments) is ~500 short launches and therefore launch-bound; a
        graph per (batch size, dtype, parameter storage) replays them without per-launch host work.  BatchNorm eval
        coefficients are computed inside the graph from the live running statistics; the fragment-ordered weight
        packs are refreshed (outside the graph, only when the weights changed) into the buffers the graph reads."""
        m = self.model
        dev = obs.device
        if obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.float().contiguous()
        self._get_packs(T, dev)
        # (the flat tensor lists are cached while no module registered a submodule / parameter / buffer -- walking the module
        # tree costs more host time than the whole graph replay; the storage addresses themselves are checked on every call)
        from keisei_amd.training._structure import structure_fingerprint, structure_version
        tl, ver = self._tensor_lists, structure_version()
        self._tl_calls = getattr(self, "_tl_calls", 0) + 1
        # (backstop every 64 calls: an edit that goes around every hook, e.g. `m._modules.pop(name)`, changes the counts)
        if tl is not None and tl[2] == ver and self._tl_calls % 64 == 0 and structure_fingerprint(m) != tl[3]:
            tl = None
        if tl is None or tl[2] != ver:
            tl = self._tensor_lists = (list(m.buffers()), list(m.parameters()), ver, structure_fingerprint(m))
        key = (tuple(obs.shape), T, str(dev), self._pack_tkey,
               tuple(b.data_ptr() for b in tl[0]), tuple(q.data_ptr() for q in tl[1]))
        with self._graph_lock:
            ent = self._graphs.get(key)
            if ent is None:
                if len(self._graphs) >= 8:
                    self._graphs.clear()
                static_in = obs.clone()
                saved_overlap = self.fork_fc
                # KA_EVAL_GRAPH_FORK=1: keep the global-pool FC chain on the side stream inside the capture (a fork /
                # join in the graph, beside conv1); 0 = single-stream capture
                self.fork_fc = saved_overlap and os.environ.get("KA_EVAL_GRAPH_FORK", "1") != "0"
                try:
                    self.forward(static_in, False, False, T, None)            # warm-up: one-time kernel attributes etc.
                    torch.cuda.synchronize(dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        out = self.forward(static_in, False, False, T, None)[:3]
                finally:
                    self.fork_fc = saved_overlap
                ent = (graph, static_in, out)
                self._graphs[key] = ent
            graph, static_in, out = ent
            static_in.copy_(obs)
            graph.replay()
            return tuple(t.clone() for t in out)

    # ------------------------------------------------------------------ forward
    def forward(self, obs: torch.Tensor, train: bool, keep: bool, T: torch.dtype, idx: Optional[torch.Tensor] = None):
        self._in_forward = True
        m = self.model
        p = m.params
        dev = obs.device
        st = _lib.stream_ptr(dev)
        code = _lib.dtype_code(T)
        C = p.channels
        B = obs.shape[0] if idx is None else idx.shape[0]
        if obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.float().contiguous()
        packs = self._get_packs(T, dev)
        self._evalc_live = self._eval_coeffs_all(dev, st) if (not train and not keep) else None
        rows = _lib.query("ka_conv3x3_sqpart_rows", B)
        count = B * 81
        sv = _Saved()
        sv.B, sv.T, sv.train = B, T, train

        def new_act(ch):
            return torch.empty(B, 81, ch, dtype=T, device=dev)

        # ---- stem
        cin_pad = _round_up(p.obs_channels, 64)
        xin = new_act(cin_pad)
        _call("ka_obs_to_nhwc", obs, idx, xin, B, p.obs_channels, cin_pad, code, st)
        y0 = new_act(C)
        bsum = torch.empty(B, C, device=dev); sq = torch.empty(rows, C, device=dev)
        _call("ka_conv3x3_fwd", xin, packs["input_conv"][0], y0, None, None, None, 0,
              bsum if train else None, sq if train else None, B, cin_pad, C, code, st)
        sc0, sh0, mu0, is0 = self._bn_forward(m.input_bn, bsum, B, sq, rows, C, count, train, dev, st)
        x = new_act(C)
        pool = torch.empty(B, 4 * C, device=dev)        # [mean|max|std|ties]
        _call("ka_block_tail_fwd", y0, sc0, sh0, None, None, x, pool, B, C, code, st)
        sv.xin = xin
        sv.stem = (y0, sc0, sh0, mu0, is0)
        sv.blocks = []

        # ---- tower
        tower_tab = self._tower_table(T, dev, packs) if (not train and not keep and len(m.blocks) > 0) else None
        if tower_tab is not None:
            # eval mode: no tensor couples the boards, so ONE launch carries every board through all the blocks
            # (csrc/tower.hip) instead of 2 conv + 2 FC-chain + 1 tail launch per block
            blk0 = m.blocks[0]
            out = new_act(C)
            pool_out = torch.empty(B, 4 * C, device=dev)
            _call("ka_tower_eval", x, pool, out, pool_out, tower_tab, len(m.blocks), B, C, blk0.global_fc[0].out_features,
                  blk0.se_fc1.out_features, code, st)
            x, pool = out, pool_out
        # The global-pool FC chain (13 us at B = 4096) runs on the main stream, between conv1's statistics reduce and its
        # coefficient kernel.  Forked onto the side stream (KA_FC_SIDE=1, the round-1/2 arrangement) it cannot share a CU with
        # the persistent conv workgroups anyway: it either takes CUs first and delays conv1, or waits until conv1 has drained
        # (336 us per launch in the round-2 trace).  Small batches (rollout inference without the tower kernel) keep the
        # fork: their conv workgroups leave CUs free.
        fc_side = os.environ.get("KA_FC_SIDE")
        fside = self._wgrad_side(1, dev)[0] if (self.fork_fc and (fc_side == "1" or (fc_side is None and B < 512))) else None
        # KA_FC_SIDE=2 (experiment): forked BEHIND conv1 instead -- eligible when conv1 has drained, i.e. beside the statistics kernels
        flate = self._wgrad_side(1, dev)[0] if (fc_side == "2" and fside is None) else None
        main_f = torch.cuda.current_stream(dev)
        keep_x2 = (T == torch.bfloat16 and os.environ.get("KA_KEEP_X2", "0") == "1"
                   and bool(_lib.query("ka_conv3x3_fwd_keep_supported", B, C, C, code)))
        # the squeeze-excite FC chain of a block inside its tail launch (ka_block_tail_fwd_se): KA_SE_IN_TAIL=1
        se_in_tail = os.environ.get("KA_SE_IN_TAIL", "0") == "1"
        for i, blk in enumerate(m.blocks if tower_tab is None else ()):
            # g = global_fc(pool(x)) is only needed by conv2
            if fside is not None:
                ev = torch.cuda.Event(); ev.record(main_f)
                with torch.cuda.stream(fside):
                    fside.wait_event(ev)
                    sts = _lib.stream_ptr(dev)
                    _, g1, g = self._fc_chain(pool, blk.global_fc[0], blk.global_fc[2], sts, keep)
                    g_ready = torch.cuda.Event(); g_ready.record(fside)
                g.record_stream(main_f)
                if g1 is not None:
                    g1.record_stream(main_f)
            y1 = new_act(C)
            bsum1 = torch.empty(B, C, device=dev); sq1 = torch.empty(rows, C, device=dev)
            self._timed("conv3x3", "ka_conv3x3_fwd", x, packs[f"blocks.{i}.conv1"][0], y1, None, None, None, 0,
                  bsum1 if train else None, sq1 if train else None, B, C, C, code, st)
            if flate is not None:
                ev = torch.cuda.Event(); ev.record(main_f)
                with torch.cuda.stream(flate):
                    flate.wait_event(ev)
                    _, g1, g = self._fc_chain(pool, blk.global_fc[0], blk.global_fc[2], _lib.stream_ptr(dev), keep)
                    g_ready = torch.cuda.Event(); g_ready.record(flate)
                pool.record_stream(flate); g.record_stream(main_f)
                if g1 is not None:
                    g1.record_stream(main_f)
            bn1_ctx = self._bn_forward_begin(blk.bn1, bsum1, B, sq1, rows, C, count, train, dev, st)
            if flate is not None:
                pass
            elif fside is None:
                # on the main stream BEHIND conv1's statistics reduce: under SyncBatchNorm the chain (independent of bn1)
                # runs while the layer's statistics all-reduce is on the wire
                _, g1, g = self._fc_chain(pool, blk.global_fc[0], blk.global_fc[2], st, keep)
            sc1, sh1, mu1, is1 = self._bn_forward_end(bn1_ctx)
            if fside is not None or flate is not None:
                main_f.wait_event(g_ready)               # global-pool FC chain ran on the side stream beside conv1
            y2 = new_act(C)
            bsum2 = torch.empty(B, C, device=dev); sq2 = torch.empty(rows, C, device=dev)
            # x2 = relu(bn1(y1)) + g, the tensor conv2 multiplies: with KA_KEEP_X2=1 it is kept (written by the conv's own staging waves)
            # so that conv2's weight gradient reads a plain operand instead of recomputing it per tile.  Measured: weight gradients
            # -16 us per launch, conv launches +3 us, the step 98.02 -> 97.87 ms, for 6.8 GB more HBM traffic and memory: off by default
            x2 = new_act(C) if (keep and keep_x2) else None
            if x2 is not None:
                self._timed("conv3x3", "ka_conv3x3_fwd_keep", y1, packs[f"blocks.{i}.conv2"][0], y2, sc1, sh1, g, 1,
                            bsum2, sq2 if train else None, x2, B, C, C, code, st)
            else:
                self._timed("conv3x3", "ka_conv3x3_fwd", y1, packs[f"blocks.{i}.conv2"][0], y2, sc1, sh1, g, 1,
                            bsum2, sq2 if train else None, B, C, C, code, st)
            sc2, sh2, mu2, is2 = self._bn_forward(blk.bn2, bsum2, B, sq2, rows, C, count, train, dev, st)
            out = new_act(C)
            pool_out = torch.empty(B, 4 * C, device=dev)
            Hse = blk.se_fc1.weight.shape[0]
            if (se_in_tail and blk.se_fc1.weight.is_contiguous() and blk.se_fc2.weight.is_contiguous()
                    and blk.se_fc1.bias is not None and blk.se_fc2.bias is not None
                    and _lib.query("ka_block_tail_fwd_se_supported", C, Hse, code)):
                # the board's squeeze-excite chain inside the tail launch (one launch less per block)
                sqz = torch.empty(B, C, device=dev) if keep else None
                se1 = torch.empty(B, Hse, device=dev) if keep else None
                se = torch.empty(B, 2 * C, device=dev)
                _call("ka_block_tail_fwd_se", y2, sc2, sh2, bsum2, blk.se_fc1.weight, blk.se_fc1.bias, blk.se_fc2.weight,
                      blk.se_fc2.bias, x, out, pool_out, sqz, se1, se, B, C, Hse, code, st)
            else:
                sqz, se1, se = self._fc_chain(bsum2, blk.se_fc1, blk.se_fc2, st, keep, affine=(sc2, sh2, 1.0 / 81.0))
                _call("ka_block_tail_fwd", y2, sc2, sh2, se, x, out, pool_out, B, C, code, st)
            if keep:
                sv.blocks.append((x, pool, y1, sc1, sh1, mu1, is1, g1, g, y2, sc2, sh2, mu2, is2, sqz, se1, se, out, x2))
            x, pool = out, pool_out

        # ---- heads
        P = p.policy_channels
        M = B * 81
        p1 = torch.empty(M, P, device=dev)
        wp1 = m.policy_conv1.weight
        self._gemm(x, wp1, p1, None, M, P, C, C, C, P, 0, 1, st, abf=int(T == torch.bfloat16))
        nsp = max(1, min(256, (M + 2047) // 2048))
        if train:
            part = torch.empty(nsp, P, device=dev); part2 = torch.empty(nsp, P, device=dev)
            _call("ka_rows_sq_sums", p1, part, part2, M, P, nsp, st)
        else:
            part = part2 = None
        scp, shp, mup, isp = self._bn_forward(m.policy_bn1, part, nsp, part2, nsp, P, M, train, dev, st)
        p1r = torch.empty(M, P, device=dev)
        _call("ka_rows_affine_relu", p1, scp, shp, p1r, M, P, st)
        A = m.SPATIAL_MOVE_TYPES
        logits = torch.empty(B, 9, 9, A, device=dev)
        wp2 = m.policy_conv2.weight
        self._gemm(p1r, wp2, logits, m.policy_conv2.bias, M, A, P, P, P, A, 0, 1, st)
        _, v1, v = self._fc_chain(pool, m.value_fc1, m.value_fc2, st, keep)
        _, s1, s = self._fc_chain(pool, m.score_fc1, m.score_fc2, st, keep)
        if keep:
            sv.heads = (x, pool, p1, scp, shp, mup, isp, p1r, v1, s1)
        self._in_forward = False
        self._evalc_live = None
        return logits, v, s, (sv if keep else None)

    # ------------------------------------------------------------------ backward
    def _bn_backward_begin(self, bn, s1p, s2p, rows, C, count, train, dev, st):
        """stage-1 reduce of the two BatchNorm-backward sums and, under SyncBatchNorm, their asynchronous all-reduce"""
        ws = self._red_ws(C, dev)
        if train and self._sync_group(bn):
            sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
            gsums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
            _call("ka_sync_reduce", s1p, rows, s2p, rows, C, float(count), gsums, sums, ws, st)
            return (sums, gsums, self._allreduce_async(gsums), ws)
        if self.bn_one_launch and C <= 64 * 64:
            return (None, None, None, (ws, s1p, s2p, rows))                  # one launch, in _bn_backward_end
        _call("ka_pair_reduce", s1p, s2p, rows, C, None, ws, st)             # stage 1 only: partials stay in ws
        return (None, None, None, ws)

    def _bn_backward_end(self, ctx, bn, C, count, mu, istd, train, grads, prefix, dev, st):
        sums, gsums, work, ws = ctx
        dgam = torch.empty(C, device=dev); dbet = torch.empty(C, device=dev)
        k = torch.empty(3 * C, device=dev)
        if gsums is not None:
            self._allreduce_wait(work)
            _call("ka_bn_bwd_coeffs", sums, gsums, float(count), gsums[2 * C:], bn.weight, mu, istd, dgam, dbet, k, C,
                  1 if train else 0, st)
        elif isinstance(ws, tuple):
            ws, s1p, s2p, rows = ws
            _call("ka_pair_reduce_bwd_coeffs", s1p, s2p, rows, C, ws, self._red_counters(dev), float(count), bn.weight, mu, istd,
                  dgam, dbet, k, 1 if train else 0, st)
        else:
            _call("ka_bn_bwd_coeffs_parts", ws, float(count), bn.weight, mu, istd, dgam, dbet, k, C, 1 if train else 0, st)
        grads[prefix + ".weight"], grads[prefix + ".bias"] = dgam, dbet
        return k

    def _bn_backward(self, bn, s1p, s2p, rows, C, count, mu, istd, train, grads, prefix, dev, st):
        ctx = self._bn_backward_begin(bn, s1p, s2p, rows, C, count, train, dev, st)
        return self._bn_backward_end(ctx, bn, C, count, mu, istd, train, grads, prefix, dev, st)

    def backward(self, sv: _Saved, dlogits, dv, ds) -> Dict[str, torch.Tensor]:
        m = self.model
        p = m.params
        B, T, train = sv.B, sv.T, sv.train
        code = _lib.dtype_code(T)
        bf = int(T == torch.bfloat16)
        C, P, A = p.channels, p.policy_channels, m.SPATIAL_MOVE_TYPES
        x, pool, p1, scp, shp, mup, isp, p1r, v1, s1 = sv.heads
        dev = x.device
        st = _lib.stream_ptr(dev)
        packs = self._get_packs(T, dev)
        grads: Dict[str, torch.Tensor] = {}
        M = B * 81
        count = M
        self._fc_jobs = [] if os.environ.get("KA_FC_WGRAD_GROUPED", "1") != "0" else None
        # the global-pool chain's input gradients in one launch per block (transposed weight copies, refreshed once per step)
        fc_tr = None
        if len(m.blocks) > 0 and os.environ.get("KA_FC_CHAIN_BWD", "1") != "0":
            l1, l2 = m.blocks[0].global_fc[0], m.blocks[0].global_fc[2]
            if (_lib.query("ka_fc_chain_supported", l2.out_features, l2.out_features, l1.out_features, l1.in_features)
                    and all(b.global_fc[0].weight.is_contiguous() and b.global_fc[2].weight.is_contiguous() for b in m.blocks)):
                fc_tr = self._fc_transposed(dev, st)

        def new_act(ch=C):
            return torch.empty(B, 81, ch, dtype=T, device=dev)

        def eval_stats(bn):
            return bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)

        # ---- value / score heads -> dpool
        dpool = torch.zeros(B, 3 * C, device=dev)
        for d_out, hid, fc1, fc2, n1, n2 in ((dv, v1, m.value_fc1, m.value_fc2, "value_fc1", "value_fc2"),
                                              (ds, s1, m.score_fc1, m.score_fc2, "score_fc1", "score_fc2")):
            if d_out is None:
                grads[n2 + ".weight"] = torch.zeros_like(fc2.weight); grads[n2 + ".bias"] = torch.zeros_like(fc2.bias)
                grads[n1 + ".weight"] = torch.zeros_like(fc1.weight); grads[n1 + ".bias"] = torch.zeros_like(fc1.bias)
                continue
            d_out = d_out.float().contiguous()
            dh = self._linear_bwd(d_out, hid, fc2, grads, n2 + ".weight", n2 + ".bias", st)
            _call("ka_relu_mask", dh, hid, dh.numel(), st)
            self._linear_bwd(dh, pool, fc1, grads, n1 + ".weight", n1 + ".bias", st, dx_out=dpool, acc_dx=1)

        # ---- policy head
        dxc = None
        if dlogits is not None:
            dl = dlogits.float().contiguous().view(M, A)
            w2 = _FakeLinear(m.policy_conv2.weight.view(A, P), m.policy_conv2.bias)
            dp = self._linear_bwd(dl, p1r, w2, grads, "policy_conv2.weight", "policy_conv2.bias", st, defer=False)
            grads["policy_conv2.weight"] = grads["policy_conv2.weight"].view_as(m.policy_conv2.weight)
            _call("ka_rows_bn_bwd", dp, p1, scp, shp, M, P, 0, st)                # ReLU mask
            nsp = max(1, min(256, (M + 2047) // 2048))
            part = torch.empty(nsp, P, device=dev); part2 = torch.empty(nsp, P, device=dev)
            mu_p, is_p = (mup, isp) if train else eval_stats(m.policy_bn1)
            _call("ka_rows_bn_sums", dp, p1, mu_p, is_p, part, part2, M, P, nsp, st)
            k = self._bn_backward(m.policy_bn1, part, part2, nsp, P, M, mu_p, is_p, train, grads, "policy_bn1", dev, st)
            _call("ka_rows_bn_bwd", dp, p1, k, None, M, P, 1, st)
            w1 = _FakeLinear(m.policy_conv1.weight.view(P, C), None)
            dxc = new_act()
            self._linear_bwd_act(dp, x, w1, grads, "policy_conv1.weight", dxc, bf, st)
            grads["policy_conv1.weight"] = grads["policy_conv1.weight"].view_as(m.policy_conv1.weight)
        else:
            for n, t in (("policy_conv2.weight", m.policy_conv2.weight), ("policy_conv2.bias", m.policy_conv2.bias),
                         ("policy_bn1.weight", m.policy_bn1.weight), ("policy_bn1.bias", m.policy_bn1.bias),
                         ("policy_conv1.weight", m.policy_conv1.weight)):
                grads[n] = torch.zeros_like(t)
        # The block-input gradient of the block above and the backward tail of the next block meet at every block boundary:
        # one launch for the two (ka_block_dx_tail_bwd: 7 activation passes instead of 9, same results bit for bit) wherever the
        # shape allows it.  `pend` = the ka_block_dx call still owed: (dxc, dout_up, out_up, x, xpool, dpool, dx).
        # The launches form a chain (ka_block_dx_tail_bwd_du): each writes du = dx * [x > 0] -- all the residual branch below
        # needs -- instead of dx, and adds the du from above as it is, so the block above's output is not read (6 passes).
        Hse = m.blocks[0].se_fc1.weight.shape[0] if len(m.blocks) > 0 else 0
        fuse_dx = (len(m.blocks) > 0 and os.environ.get("KA_DX_TAIL", "1") != "0"
                   and bool(_lib.query("ka_block_dx_tail_bwd_supported", C, Hse, code)))
        # ... and without dz (ka_block_dx_tail_bwd_du_gate): dz = du * gate[b, c] + add[b, c] is formed inside the conv2 data
        # gradient's input transform (ka_conv3x3_dgrad_fused_gated) from the du the launch writes anyway -- 5 passes, and dz is
        # no longer rounded to bf16 on its way into BatchNorm's backward.  KA_TAIL_GATE=0: the dz form.
        gate_dz = (fuse_dx and T == torch.bfloat16 and os.environ.get("KA_TAIL_GATE", "1") != "0"
                   and bool(_lib.query("ka_conv3x3_dgrad_gated_supported", B, C, C, code, 1)))
        pend = (dxc, None, None, x, pool, dpool, new_act())
        dout = pend[6]
        if not fuse_dx:
            _call("ka_block_dx", *pend, B, C, code, st)
            pend = None

        s1p = torch.empty(B, C, device=dev); s2p = torch.empty(B, C, device=dev)
        # overlapped wgrad leaves a quarter of the CUs to the concurrent main-stream kernels (measured best: 192 of 256)
        twg = 192 if self.overlap_wgrad else 0
        nsplit = _lib.query("ka_wgrad_splits", B, C, C, twg)
        nslab = max(nsplit * 9 * C * C, _lib.query("ka_wgrad_splits", B, 64, C, 0) * 9 * C * 64)
        main = torch.cuda.current_stream(dev)
        side, slab = self._wgrad_side(nslab, dev)
        if not self.overlap_wgrad:
            side = None

        # gradient exchange driven from inside the pass (hip/grad_reducer.py): conv weight gradients are written straight
        # into flat bucket buffers whose all-reduce is issued as soon as the bucket's last wgrad kernel is queued
        red = self.grad_reducer
        per_block = 2 * 9 * C * C
        blocks_per_bucket = max(1, red.bucket_bytes // (4 * per_block)) if red is not None else 0
        bucket, bucket_off, bucket_left = None, 0, 0

        def conv_grad(like, i, last):
            """dW tensor of a tower convolution of block i (`last`: second and final one of the block)"""
            nonlocal bucket, bucket_off, bucket_left
            if red is None:
                return torch.empty_like(like)
            if bucket is None:
                bucket_left = min(blocks_per_bucket, i + 1)
                bucket = torch.empty(bucket_left * per_block, device=dev)
                bucket_off = 0
            n = like.numel()
            view = bucket[bucket_off:bucket_off + n].view_as(like)
            bucket_off += n
            return view

        def conv_grads_done(i):
            """both weight gradients of block i are queued: close the bucket when it is full"""
            nonlocal bucket, bucket_left
            if red is None:
                return
            bucket_left -= 1
            if bucket_left == 0:
                ev = torch.cuda.Event()
                ev.record(side if side is not None else main)
                red.launch(bucket, f"conv[{i}:{i + bucket.numel() // per_block}]", ev)
                bucket = None

        fc_bwd_side = None
        if os.environ.get("KA_FC_BWD_SIDE", "0") == "1" and fc_tr is not None and not self.overlap_wgrad:
            if self._fc_side is None or self._fc_side.device != dev:
                self._fc_side = torch.cuda.Stream(dev)
            fc_bwd_side = self._fc_side
        self._gpool_done = None

        def gpool_join():
            """the chain launch of the block above (side stream) must be done before its dpool / dg1 are read"""
            if self._gpool_done is not None:
                main.wait_event(self._gpool_done)
                self._gpool_done = None

        # ---- tower, last block first
        for i in range(len(sv.blocks) - 1, -1, -1):
            blk = m.blocks[i]
            (bx, bpool, y1, sc1, sh1, mu1, is1, g1, g, y2, sc2, sh2, mu2, is2, sqz, se1, se, out, x2) = sv.blocks[i]
            pre = f"blocks.{i}."
            if not train:
                mu1, is1 = eval_stats(blk.bn1); mu2, is2 = eval_stats(blk.bn2)
            dse = torch.empty(B, 2 * C, device=dev)
            H = blk.se_fc1.weight.shape[0]
            gpool_join()
            gate_add = None
            if pend is not None and gate_dz:
                dse1 = torch.empty(B, H, device=dev)
                gate_add = torch.empty(2, B, C, device=dev)
                dz = pend[6]                                          # the du this launch writes; dz = du * gate + add
                _call("ka_block_dx_tail_bwd_du_gate", pend[0], pend[1], *pend[3:], y2, sc2, sh2, se, se1, blk.se_fc2.weight,
                      blk.se_fc1.weight, mu2, is2, gate_add[0], gate_add[1], dse, dse1, s1p, s2p, B, C, H, code, st)
                self._linear_bwd(dse, se1, blk.se_fc2, grads, pre + "se_fc2.weight", pre + "se_fc2.bias", st, need_dx=False)
                self._linear_bwd(dse1, sqz, blk.se_fc1, grads, pre + "se_fc1.weight", pre + "se_fc1.bias", st, need_dx=False)
            elif pend is not None:
                # (pend[3], the input of the block above, IS this block's output `out`)
                dse1 = torch.empty(B, H, device=dev)
                dz = new_act()
                _call("ka_block_dx_tail_bwd_du", pend[0], pend[1], *pend[3:], y2, sc2, sh2, se, se1, blk.se_fc2.weight,
                      blk.se_fc1.weight, mu2, is2, dz, dse, dse1, s1p, s2p, B, C, H, code, st)
                self._linear_bwd(dse, se1, blk.se_fc2, grads, pre + "se_fc2.weight", pre + "se_fc2.bias", st, need_dx=False)
                self._linear_bwd(dse1, sqz, blk.se_fc1, grads, pre + "se_fc1.weight", pre + "se_fc1.bias", st, need_dx=False)
            elif _lib.query("ka_tail_bwd_fused_supported", C, H, code):
                # one read of dout / out / y2: SE-gate reductions, the per-board FC chain backward and dz in one kernel
                dse1 = torch.empty(B, H, device=dev)
                dz = new_act()
                _call("ka_tail_bwd_fused", dout, out, y2, sc2, sh2, se, se1, blk.se_fc2.weight, blk.se_fc1.weight, mu2, is2,
                      dz, dse, dse1, s1p, s2p, B, C, H, code, st)
                self._linear_bwd(dse, se1, blk.se_fc2, grads, pre + "se_fc2.weight", pre + "se_fc2.bias", st, need_dx=False)
                self._linear_bwd(dse1, sqz, blk.se_fc1, grads, pre + "se_fc1.weight", pre + "se_fc1.bias", st, need_dx=False)
            else:
                dz = new_act()
                _call("ka_tail_bwd_reduce", dout, out, y2, sc2, sh2, se, dse, B, C, code, st)
                dse1 = self._linear_bwd(dse, se1, blk.se_fc2, grads, pre + "se_fc2.weight", pre + "se_fc2.bias", st)
                _call("ka_relu_mask", dse1, se1, dse1.numel(), st)
                dsq = self._linear_bwd(dse1, sqz, blk.se_fc1, grads, pre + "se_fc1.weight", pre + "se_fc1.bias", st)
                _call("ka_tail_bwd_dz", dout, out, y2, se, dsq, mu2, is2, dz, s1p, s2p, B, C, code, st)
            k2 = self._bn_backward(blk.bn2, s1p, s2p, B, C, count, mu2, is2, train, grads, pre + "bn2", dev, st)
            dg = torch.empty(B, C, device=dev)
            if T == torch.bfloat16:
                # bf16: both BatchNorm-backward "apply" passes and the ReLU/BN1-backward reduce pass are fused into the
                # two data-gradient convolutions (input transform / epilogue); dy2 / dy1 are written for the wgrads
                rows = _lib.query("ka_conv3x3_sqpart_rows", B)
                dy2, dh = new_act(), new_act()
                ep1 = torch.empty(rows, C, device=dev); ep2 = torch.empty(rows, C, device=dev)
                if gate_add is not None:
                    self._timed("conv3x3", "ka_conv3x3_dgrad_fused_gated", dz, gate_add, y2, k2, dy2, packs[pre + "conv2"][1], dh, dg,
                                y1, sc1, sh1, mu1, is1, ep1, ep2, B, C, C, code, st)
                else:
                    self._timed("conv3x3", "ka_conv3x3_dgrad_fused", dz, y2, k2, dy2, packs[pre + "conv2"][1], dh, dg,
                                y1, sc1, sh1, mu1, is1, ep1, ep2, B, C, C, code, st)
                dW2 = conv_grad(blk.conv2.weight, i, False)
                if x2 is not None:       # the forward kept conv2's input: a plain operand
                    self._wgrad_launch(side, main, (dy2, dW2), dy2, x2, None, None, None, 0, slab, dW2, B, C, C, C, 0, twg, code)
                else:
                    self._wgrad_launch(side, main, (dy2, dW2), dy2, y1, sc1, sh1, g, 1, slab, dW2, B, C, C, C, 0, twg, code)
                grads[pre + "conv2.weight"] = dW2
                # the global-pool chain's backward (independent of bn1's coefficients) runs under bn1's statistics all-reduce
                gside = None
                if fc_bwd_side is not None:
                    # (behind the weight-gradient launch: eligible when that has drained, i.e. beside the slab reduce and the
                    #  statistics kernels, which leave the CUs to it)
                    gev = torch.cuda.Event(); gev.record(main)
                    gside = (fc_bwd_side, gev)
                bn1_ctx = self._bn_backward_begin(blk.bn1, ep1, ep2, rows, C, count, train, dev, st)
                dpool_x = self._gpool_bwd(i, blk, dg, g1, bpool, grads, pre, st, fc_tr, gside)
                k1 = self._bn_backward_end(bn1_ctx, blk.bn1, C, count, mu1, is1, train, grads, pre + "bn1", dev, st)
                dy1, dxc = new_act(), new_act()
                self._timed("conv3x3", "ka_conv3x3_dgrad_fused", dh, y1, k1, dy1, packs[pre + "conv1"][1], dxc, None,
                            None, None, None, None, None, None, None, B, C, C, code, st)
                dW1 = conv_grad(blk.conv1.weight, i, True)
                self._wgrad_launch(side, main, (dy1, dW1), dy1, bx, None, None, None, 0, slab, dW1, B, C, C, C, 0, twg, code)
                grads[pre + "conv1.weight"] = dW1
                conv_grads_done(i)
                dx = new_act()
            else:
                _call("ka_bn_bwd_apply", dz, y2, k2, dz, B, C, code, st)                         # dz -> dy2 in place
                dh = new_act()
                self._timed("conv3x3", "ka_conv3x3_fwd", dz, packs[pre + "conv2"][1], dh, None, None, None, 0, dg, None, B, C, C, code, st)
                dW2 = conv_grad(blk.conv2.weight, i, False)
                self._wgrad_launch(side, main, (dz, dW2), dz, y1, sc1, sh1, g, 1, slab, dW2, B, C, C, C, 0, twg, code)
                grads[pre + "conv2.weight"] = dW2
                dpool_x = self._gpool_bwd(i, blk, dg, g1, bpool, grads, pre, st, fc_tr)
                _call("ka_relu_bn_bwd_reduce", dh, y1, sc1, sh1, mu1, is1, dh, s1p, s2p, B, C, code, st)   # dh -> da1 in place
                k1 = self._bn_backward(blk.bn1, s1p, s2p, B, C, count, mu1, is1, train, grads, pre + "bn1", dev, st)
                _call("ka_bn_bwd_apply", dh, y1, k1, dh, B, C, code, st)                         # -> dy1 in place
                dxc = new_act() if side is not None else dz       # dz / dh may still be read by the side stream
                self._timed("conv3x3", "ka_conv3x3_fwd", dh, packs[pre + "conv1"][1], dxc, None, None, None, 0, None, None, B, C, C, code, st)
                dW1 = conv_grad(blk.conv1.weight, i, True)
                self._wgrad_launch(side, main, (dh, dW1), dh, bx, None, None, None, 0, slab, dW1, B, C, C, C, 0, twg, code)
                grads[pre + "conv1.weight"] = dW1
                conv_grads_done(i)
                dx = new_act() if side is not None else dh
            if fuse_dx and i > 0:
                pend = (dxc, dout, out, bx, bpool, dpool_x, dx)        # joins the next block's tail
            else:
                gpool_join()
                _call("ka_block_dx", dxc, dout, out, bx, bpool, dpool_x, dx, B, C, code, st)
                pend = None
            dout = dx

        # ---- stem
        y0, sc0, sh0, mu0, is0 = sv.stem
        if not train:
            mu0, is0 = eval_stats(m.input_bn)
        _call("ka_relu_bn_bwd_reduce", dout, y0, sc0, sh0, mu0, is0, dout, s1p, s2p, B, C, code, st)
        k0 = self._bn_backward(m.input_bn, s1p, s2p, B, C, count, mu0, is0, train, grads, "input_bn", dev, st)
        _call("ka_bn_bwd_apply", dout, y0, k0, dout, B, C, code, st)
        dW0 = torch.empty_like(m.input_conv.weight)
        cin_pad = sv.xin.shape[2]
        # on the wgrad stream as well: the partial-slab buffer is shared with the tower wgrads still running there
        self._wgrad_launch(side, main, (dout, dW0), dout, sv.xin, None, None, None, 0, slab, dW0, B, cin_pad,
                           p.obs_channels, C, 0, 0, code)
        grads["input_conv.weight"] = dW0
        fc_flat = self._flush_fc_jobs(grads, dev, st)
        if side is not None:
            main.wait_stream(side)          # every dW is complete before autograd hands the gradients on
        if red is not None:
            # the rest: the FC gradients already share one flat buffer; BatchNorm / head / stem gradients are coalesced
            in_flat = set()
            if fc_flat is not None:
                lo, hi = fc_flat.data_ptr(), fc_flat.data_ptr() + 4 * fc_flat.numel()
                in_flat = {n for n, t in grads.items() if lo <= t.data_ptr() < hi}
                red.launch(fc_flat, "fc", red._event_now(fc_flat))
            small_names = [n for n, t in grads.items()
                           if n not in in_flat and not (n.startswith("blocks.") and n.endswith(("conv1.weight", "conv2.weight")))]
            for n in small_names:
                red.add_small(grads[n])
            for n, v in zip(small_names, red.finish()):
                grads[n] = v                    # views of the reduced packed buffer (no per-tensor copy back)
        return grads

    def _linear_bwd_act(self, dy, x_act, lin, grads, wname, dx_act, bf, st):
        """policy conv1x1 backward where the input / its gradient are activation tensors (B*81, C) of dtype T."""
        M, N = dy.shape
        K = lin.weight.shape[1]
        dev = dy.device
        ns = max(1, min(_FC_SPLITS, (M + 1023) // 1024))
        dW = torch.empty(N, K, device=dev)
        slab = self._scratch_f32(ns * N * K, dev)
        self._gemm(dy, x_act, slab, None, N, K, M, N, K, K, 1, 0, st, bbf=bf, ns=ns)
        _call("ka_reduce_slabs", slab, dW, ns, N * K, 0, st)
        grads[wname] = dW
        self._gemm(dy, lin.weight, dx_act, None, M, K, N, N, K, K, 0, 0, st, cbf=bf)


class _FakeLinear:
    """weight/bias pair with nn.Linear's attribute names (1x1 convolutions viewed as linear maps)."""
    __slots__ = ("weight", "bias")

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


class _SEResNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: SEResNetEngine, obs, idx, train, keep, T, names, *params):
        logits, v, s, saved = engine.forward(obs, train, keep, T, idx)
        ctx.engine, ctx.saved, ctx.names = engine, saved, names
        ctx.set_materialize_grads(False)
        return logits, v, s

    @staticmethod
    def backward(ctx, dlogits, dv, ds):
        if ctx.saved is None:
            raise RuntimeError("SE-ResNet HIP backward called without saved activations")
        grads = ctx.engine.backward(ctx.saved, dlogits, dv, ds)
        ctx.saved = None
        return (None, None, None, None, None, None, None, *[grads.get(n) for n in ctx.names])


def run_model(model: nn.Module, obs: torch.Tensor, idx: Optional[torch.Tensor] = None):
    """Forward of SEResNetModel on a CUDA/HIP device.  Returns (policy (B,9,9,139), value (B,3), score (B,1))."""
    engine = getattr(model, "_hip_engine", None)
    if engine is None:
        engine = SEResNetEngine(model)
        object.__setattr__(model, "_hip_engine", engine)
    bf16 = False
    if model._amp_enabled and model._amp_dtype == torch.bfloat16:
        bf16 = True
    elif torch.is_autocast_enabled("cuda"):
        if torch.get_autocast_dtype("cuda") == torch.bfloat16:
            bf16 = True
        else:
            raise _lib.KeiseiHipError("the HIP path supports fp32 and bf16 autocast only (fp16 requested)")
    elif model._amp_enabled:
        raise _lib.KeiseiHipError("the HIP path supports fp32 and bf16 autocast only")
    T = torch.bfloat16 if bf16 else torch.float32
    if (not torch.is_grad_enabled() and not model.training and idx is None and obs.shape[0] <= 2048
            and os.environ.get("KA_EVAL_GRAPH", "1") != "0" and not torch.cuda.is_current_stream_capturing()):
        with torch.autocast("cuda", enabled=False):      # (the rollout step: no walk over the parameters)
            return engine.forward_eval_graphed(obs, T)
    named = [(n, p) for n, p in model.named_parameters()]
    keep = torch.is_grad_enabled() and any(p.requires_grad for _, p in named)
    names = tuple(n for n, _ in named)
    with torch.autocast("cuda", enabled=False):
        return _SEResNetFunction.apply(engine, obs, idx, model.training, keep, T, names, *[p for _, p in named])
