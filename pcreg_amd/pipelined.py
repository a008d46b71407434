"""Two (or more) registrations in flight per GPU: the step's launch-sized kernels of one registration run beside the big
kernels of another.

One registration is a chain of ~14 launches of which only two or three fill the chip (the search, the two scoring passes); the
rest -- seeding, finalize, the filters, the sample fits, the refits, the selection -- are latency-sized and do not shrink when the
model is sharded over more GPUs.  The metric is throughput, registrations are independent (one model, many
surfaces: completeExperimentFast.m:131-149), so `PipelinedRegistration` keeps K `RegistrationPipeline` lanes, each with its own
HIP stream, workspaces and result buffers, and deals the submitted surfaces to them round-robin.  Nothing synchronises until
`results()`.  With a process group every lane issues its three collectives in host program order -- the same order on every rank
-- so the group's collective sequence is identical on all ranks whatever the lanes' kernels do on the device."""
from __future__ import annotations

import torch

from .device import RegistrationPipeline


class PipelinedRegistration:
    def __init__(self, Q: int, M_local: int, lanes: int = 2, m_lo: int = 0, M_total: int | None = None, group=None,
                 device: torch.device | None = None, replica: bool = False):
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.lanes = [RegistrationPipeline(Q, M_local, m_lo=m_lo, M_total=M_total, group=group, device=self.dev, replica=replica)
                      for _ in range(max(1, lanes))]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in self.lanes]
        self._next = 0
        self._pending = []          # (lane index) in submission order

    def submit(self, q_soa, model, thr_abs: float, max_ratio: float, coef: dict, unique: bool = True, seed: int = 0,
               inputs_ready: bool = False) -> int:
        """Enqueue one registration (search, filters, RANSAC with the hypotheses split over the group's ranks) on the next lane;
        returns the lane.  The lane's previous result must have been fetched (results()) before it is reused more than once:
        submit() reuses lane buffers in round-robin order, so at most `lanes` registrations are in flight."""
        k = self._next
        self._next = (k + 1) % len(self.lanes)
        lane, st = self.lanes[k], self.streams[k]
        if not inputs_ready:
            st.wait_stream(torch.cuda.current_stream(self.dev))      # inputs produced on the caller's stream
        with torch.cuda.stream(st):
            lane.search_local(q_soa, model)
            lane.match_after_search(q_soa, model, thr_abs, max_ratio, unique=unique)
            lane.ransac_sharded(coef, seed=seed)
        self._pending.append(k)
        return k

    def wait(self) -> None:
        for st in self.streams:
            torch.cuda.current_stream(self.dev).wait_stream(st)

    def results(self) -> list:
        """The results of the registrations submitted since the last call, in submission order (synchronises).  Only the LAST
        result of each lane is still in its buffers: call this at least every `lanes` submissions to see them all."""
        out = []
        seen = set()
        for k in reversed(self._pending):
            if k in seen:
                continue
            seen.add(k)
            with torch.cuda.stream(self.streams[k]):
                r = self.lanes[k].fetch_result()
                r["n_pairs"] = int(self.lanes[k].n_pairs.item())
            out.append((k, r))
        self._pending = []
        return [r for _, r in reversed(out)]
