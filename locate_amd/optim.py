"""Nadam (reference libs/nadam.py:5-89) as one fused multi-tensor HIP launch pair.

Same constructor signature and state semantics as the reference optimizer (per-parameter `step`,
`m_schedule`, `exp_avg`, `exp_avg_sq`; parameters whose `.grad` is None are skipped), but the schedule state
lives on the device so that `step()` can be captured in a hipGraph together with the backward pass."""
import struct

import torch

from ._lib import check, lib


class Nadam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, schedule_decay=4e-3):
        if weight_decay != 0:
            raise NotImplementedError("weight_decay is never used by the reference training loop")
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

    def _table(self, plist):
        """Device tables for one set of (parameter, gradient) buffers; cached on the buffer addresses."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        tab = self._tables.get(key)
        if tab is not None:
            return tab
        L = lib()
        assert L.locate_nadam_tensor_record_bytes() == 48
        chunk = L.locate_nadam_chunk_elems()
        rec = bytearray()
        chunks = []
        for i, p in enumerate(plist):
            st = self._state_for(p)
            rec += struct.pack("<5Qq", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                               st["sched"].data_ptr(), p.numel())
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
                                          torch.cuda.current_stream().cuda_stream), "locate_nadam_step")
            # the kernel writes through raw pointers: tell autograd / the packed-panel cache that the weights changed
            torch._C._increment_version(plist)
        return loss
