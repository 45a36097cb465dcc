"""Nadam (reference libs/nadam.py:5-89) as one fused multi-tensor HIP launch pair.

Same constructor signature and state semantics as the reference optimizer (per-parameter `step`,
`m_schedule`, `exp_avg`, `exp_avg_sq`; parameters whose `.grad` is None are skipped), but the schedule state
lives on the device so that `step()` can be captured in a hipGraph together with the backward pass."""
import struct

import torch

from ._lib import check, lib


class Nadam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, schedule_decay=4e-3):
        if weight_decay < 0:
            raise ValueError("weight_decay must be >= 0")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, schedule_decay=schedule_decay)
        super().__init__(params, defaults)
        self._tables = {}

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["sched"] = torch.tensor([0.0, 1.0], dtype=torch.float64, device=p.device)   # step, m_schedule
        return st

    # ---- checkpointing --------------------------------------------------------------------------------------------
    # torch.optim.Optimizer.load_state_dict casts every floating-point state tensor to the parameter's dtype, which would
    # turn the float64 (step, m_schedule) pair into float32 - the schedule kernel reads it as two doubles.  The state is
    # restored to float64 (from the checkpoint's own values, which are exact: a step count and a product of a few factors)
    # and the device tables, which hold the OLD state buffers' addresses, are dropped.
    def load_state_dict(self, state_dict):
        sched = {}
        for idx, st in state_dict.get("state", {}).items():
            if "sched" in st:
                sched[idx] = st["sched"].detach().to("cpu", torch.float64).clone()
        super().load_state_dict(state_dict)
        order = [p for group in self.param_groups for p in group["params"]]
        ids = [i for group in state_dict["param_groups"] for i in group["params"]]
        for idx, p in zip(ids, order):
            st = self.state.get(p)
            if st and idx in sched:
                st["sched"] = sched[idx].to(p.device)
        self._restore_invariants()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._restore_invariants()

    def _restore_invariants(self):
        self._tables = {}
        for p, st in self.state.items():
            if "sched" in st and st["sched"].dtype != torch.float64:
                st["sched"] = st["sched"].to(torch.float64)
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st and not st[k].is_contiguous():
                    st[k] = st[k].contiguous()

    def _absmax_words(self, p):
        """Device words that receive the largest magnitude of p after every step (the fp16-piece weight panels' scale: their
        re-packing then needs no pass of its own over the weights, ops.refresh_panels).  Only weights of 2 or more dimensions
        can own panels; the words are valid for the parameter version stamped in step()."""
        hold = p.__dict__.get("_locate_wmax")
        if hold is None or hold[0].device != p.device:
            hold = [torch.zeros(lib().locate_absmax_words(), dtype=torch.int32, device=p.device), -1]
            p.__dict__["_locate_wmax"] = hold
        return hold[0]

    def _table(self, plist):
        """Device tables for one set of (parameter, gradient, state) buffers; cached on ALL the addresses a record holds."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self._absmax_words(p).data_ptr())
                    + tuple(self._state_for(p)[k].data_ptr() for k in ("exp_avg", "exp_avg_sq", "sched")) for p in plist)
        tab = self._tables.get(key)
        if tab is not None:
            return tab
        L = lib()
        assert L.locate_nadam_tensor_record_bytes() == 56
        chunk = L.locate_nadam_chunk_elems()
        rec = bytearray()
        chunks = []
        for i, p in enumerate(plist):
            st = self._state_for(p)
            rec += struct.pack("<5QqQ", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                               st["sched"].data_ptr(), p.numel(), self._absmax_words(p).data_ptr())
            chunks.extend((i, c) for c in range((p.numel() + chunk - 1) // chunk))
        dev = plist[0].device
        # pinned staging + async copies: legal inside a hipGraph capture (the memcpy nodes re-read these host
        # buffers on every replay, so they are kept alive with the table)
        t_host = torch.frombuffer(rec, dtype=torch.uint8).clone().pin_memory()
        c_host = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).pin_memory()
        t_dev = t_host.to(dev, non_blocking=True)
        c_dev = c_host.to(dev, non_blocking=True)
        coef = torch.empty(len(plist) * 4, dtype=torch.float32, device=dev)
        tab = (t_dev, c_dev, coef, len(plist), len(chunks), t_host, c_host)
        if len(self._tables) > 8:
            self._tables.clear()
        self._tables[key] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            plist = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise TypeError("Nadam: parameters must be contiguous float32 tensors on the MI355X")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                plist.append(p)
            if not plist:
                continue
            t_dev, c_dev, coef, n_t, n_c = self._table(plist)[:5]
            b1, b2 = group["betas"]
            check(lib().locate_nadam_step(t_dev.data_ptr(), coef.data_ptr(), c_dev.data_ptr(), n_t, n_c, float(group["lr"]),
                                          float(b1), float(b2), float(group["eps"]), float(group["schedule_decay"]),
                                          float(group["weight_decay"]),
                                          torch.cuda.current_stream().cuda_stream), "locate_nadam_step")
            # the kernel writes through raw pointers: tell autograd / the packed-panel cache that the weights changed
            torch._C._increment_version(plist)
            for p in plist:
                p.__dict__["_locate_wmax"][1] = p._version        # the words hold the maximum of THIS version of the weights
        return loss
