"""Host side of the fused GradScaler.unscale_ + clip_grad_norm_ + Adam.step + GradScaler.update launch
(``ka_clip_adam_step``; reference sequence katago_ppo.py:926-933 and sl/trainer.py:163-169), shared by the PPO update and
the supervised trainer.  The optimiser state tensors stay ``torch.optim.Adam``'s own, so checkpoints interchange."""
from __future__ import annotations

import torch

from keisei_amd import _lib


class FusedAdamMixin:
    """Expects ``self.optimizer`` (a plain single-group ``torch.optim.Adam``) and ``self._hip_state`` (a dict)."""

    def _fused_optimizer_ok(self) -> bool:
        opt = self.optimizer
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            return False
        return all(q.dtype == torch.float32 and q.is_contiguous() for q in g["params"])

    def _adam_tables(self, device):
        """(Re)build the multi-tensor descriptor table for the fused clip+Adam kernel.  State tensors are the
        optimiser's own (``exp_avg`` / ``exp_avg_sq``), so checkpoints stay interchangeable with torch.optim.Adam."""
        st = self._hip_state
        opt = self.optimizer
        params = [q for q in opt.param_groups[0]["params"] if q.requires_grad]
        chunk = _lib.query("ka_adam_chunk")
        step0 = 0.0
        for q in params:
            s = opt.state[q]
            if "exp_avg" not in s:
                s["step"] = torch.tensor(0.0, dtype=torch.float32)
                s["exp_avg"] = torch.zeros_like(q, memory_format=torch.preserve_format)
                s["exp_avg_sq"] = torch.zeros_like(q, memory_format=torch.preserve_format)
            step0 = max(step0, float(s["step"]))
        key = (id(opt), tuple(q.data_ptr() for q in params))
        if st.get("key") != key:
            blk_t, blk_o = [], []
            for i, q in enumerate(params):
                for off in range(0, q.numel(), chunk):
                    blk_t.append(i); blk_o.append(off)
            st["key"] = key
            st["params"] = params
            st["blk_t"] = torch.tensor(blk_t, dtype=torch.int32, device=device)
            st["blk_o"] = torch.tensor(blk_o, dtype=torch.int64, device=device)
            st["nblocks"] = len(blk_t)
            st["partial"] = torch.empty(len(blk_t), dtype=torch.float64, device=device)
            st["tab_host"] = [torch.empty(len(params) * 5, dtype=torch.int64).pin_memory() for _ in range(2)]
            st["tab_dev"] = [torch.empty(len(params) * 5, dtype=torch.int64, device=device) for _ in range(2)]
            st["tab_evt"] = [None, None]
            st["flip"] = 0
            st["ctl"] = torch.zeros(4, device=device)
        st["step_dev"] = torch.tensor([step0], device=device)
        return st

    def _upload_table(self, st, device):
        """pointer table for this step's (freshly allocated) gradient tensors; double-buffered pinned upload,
        so the host never blocks on the stream."""
        i = st["flip"]
        st["flip"] ^= 1
        if st["tab_evt"][i] is not None:
            st["tab_evt"][i].synchronize()
        host = st["tab_host"][i]
        opt = self.optimizer
        rows = []
        for q in st["params"]:
            s = opt.state[q]
            g = q.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                raise _lib.KeiseiHipError("fused optimiser step needs a contiguous fp32 gradient for every parameter")
            rows += [q.data_ptr(), g.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), q.numel()]
        host.copy_(torch.tensor(rows, dtype=torch.int64))
        st["tab_dev"][i].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        st["tab_evt"][i] = ev
        return st["tab_dev"][i]

