"""The whole optimisation step of the hot path replayed from ONE captured hipGraph:

    random draws (torch device RNG, graph-safe) -> mipnerf_train_step (forward of both levels, loss incl. distloss, backward)
    -> Adam with the MipLRDecay schedule evaluated on the device (mipnerf_adam_step_scheduled) -> re-pack of the MFMA
    weight streams from the updated fp32 master parameters (mipnerf_set_params)

i.e. MipNeRFSystem.training_step + loss.backward() + optimizer.step() + scheduler.step() of the reference's loop
(nerf_system.py:70-76, 95-121) with no per-step host scalar: the step counter, learning rate and bias corrections live in
device memory.  With world_size > 1 the step is two graphs around one eager RCCL all-reduce of the flat gradient buffer
(SUM; the 1/world mean is folded into the Adam kernel): graph A = draws + forward + backward, graph B = Adam + re-pack.

Inputs live in static buffers (`self.rays`, `self.gt`): write the next batch there (copy_ or device-side ray generation
straight into them) and call the object.  Returns the static scalars tensor [6] = loss, mse_c, mse_f, distloss_c,
distloss_f, psnr_fine (valid until the next call).
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from .rays import Rays


class GraphedTrainStep:
    def __init__(self, system, optimizer, num_rays: int, device: torch.device, use_graph: bool = True):
        from .optim import FlatAdam
        if not isinstance(optimizer, FlatAdam):
            raise TypeError("GraphedTrainStep needs the flat optimiser (system.fused_adam = True)")
        model = system.mip_nerf
        if model.precision != L.PREC_BF16:
            raise NotImplementedError("GraphedTrainStep is the bf16 native training path")
        self.system, self.model, self.opt = system, model, optimizer
        self.B, self.N, self.dev = int(num_rays), model.num_samples, device
        self.use_graph = bool(use_graph)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the collective form (graph A, all-reduce on the process group's communicator, graph B) is what every world > 1 run
        # uses; MIPNERF_FORCE_COLLECTIVE_PATH=1 takes it at world 1 too when a process group exists (a 1-rank RCCL
        # communicator is legal), so a single-GPU box exercises the two-graph + RCCL sequence end to end
        self.collective = self.world > 1 or (os.environ.get("MIPNERF_FORCE_COLLECTIVE_PATH") == "1"
                                             and dist.is_available() and dist.is_initialized())
        self.capture_error = None
        self.time_allreduce = False        # True: a HIP event pair around every all-reduce (allreduce_stats)
        self._ar_events = []
        optimizer.grad_scale = 1.0 / self.world
        optimizer._graph_driven = True      # its steps are graph replays: lr_schedule.DeviceMipLRDecay.step reads this
        self.rays = Rays(*[torch.zeros(self.B, k, device=device) for k in (3, 3, 3, 1, 1, 1, 1)])
        self.rays.directions[:, 2] = 1.0
        self.rays.viewdirs[:, 2] = 1.0
        self.rays.radii.fill_(1e-3)
        self.rays.lossmult.fill_(1.0)
        self.rays.near.fill_(2.0)
        self.rays.far.fill_(6.0)
        self.gt = torch.zeros(self.B, 3, device=device)
        self.scalars = torch.zeros(6, device=device)
        self.randomized = bool(system.train_randomized)
        self.white = bool(system.white_bkgd)
        self.hp = system.hparams
        self._graphs = None
        self._work = None
        mlp = model.mlp
        self.ctx = mlp.native(device)
        need = int(L.lib().mipnerf_train_workspace_bytes(self.ctx.handle, self.B))
        self.ws = self.ctx.scratch("train_step", need)
        mlp.gather_foreign_grads()
        self._rp = L.RaysPtrs(*[t.data_ptr() for t in self.rays])
        ptrs = [p.data_ptr() for p in mlp.ordered_params()]
        self._params = (C.c_void_p * len(ptrs))(*ptrs)

    # ---- the two halves ------------------------------------------------------------------------------------------------
    def _fwd_bwd(self):
        m, B, N = self.model, self.B, self.N
        t_rand = u_rand = dz = None
        if self.randomized:
            draws = torch.rand(2, B, N + 1, device=self.dev)        # one launch for both draws
            t_rand, u_rand = draws[0], draws[1]                     # mip.py:159 / mip.py:201
            if m.density_noise > 0:
                dz = torch.randn(m.num_levels, B, N, device=self.dev)   # mip_nerf.py:232-233
        flags = L.FLAG_WHITE_BKGD if self.white else 0
        L.check(L.lib().mipnerf_train_step(
            self.ctx.handle, B, C.byref(self._rp), self.gt.data_ptr(), None if t_rand is None else t_rand.data_ptr(),
            None if u_rand is None else u_rand.data_ptr(), None if dz is None else dz.data_ptr(), flags,
            float(self.hp['loss.coarse_loss_mult']), 0.01, int(bool(self.hp['loss.disable_multiscale_loss'])),
            self.ws.data_ptr(), self.ws.numel(), m.mlp._flat_grad.data_ptr(), 0, self.scalars.data_ptr(), None, ops._stream()),
            "train_step")

    def _update(self):
        self.opt.launch_scheduled(count_on_host=False)
        # the packed bf16 / fp32 weight streams follow the master parameters inside the same graph
        L.check(L.lib().mipnerf_set_params(self.ctx.handle, self._params, ops._stream()), "mipnerf_set_params")

    def _capture(self):
        try:
            self._capture_impl()
        except RuntimeError as e:      # e.g. a collective library thread touching the device during a global-mode capture
            import sys
            print(f"[mipnerf_pl_amd] hipGraph capture of the training step failed ({e}); running it eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            self.use_graph, self._graphs, self.capture_error = False, None, f"{type(e).__name__}: {e}"

    def _capture_impl(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        snap = self._snapshot()
        with torch.cuda.stream(s):          # warm-up on a side stream (lazy initialisation of kernels / generators)
            self._fwd_bwd()
            self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._restore(snap)                 # the warm-up must not count as a training step
        if not self.collective:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._fwd_bwd()
                self._update()
            self._graphs = (g,)
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                self._fwd_bwd()
            with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                self._update()
            self._graphs = (ga, gb)

    def _snapshot(self):
        mlp, st = self.model.mlp, self.opt.state[self.opt._key()]
        return (mlp._flat_param.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                None if self.opt._dev_step is None else self.opt._dev_step.clone())

    def _restore(self, snap):
        mlp, st = self.model.mlp, self.opt.state[self.opt._key()]
        mlp._flat_param.copy_(snap[0])
        st["exp_avg"].copy_(snap[1])
        st["exp_avg_sq"].copy_(snap[2])
        if self.opt._dev_step is not None:
            self.opt._dev_step.fill_(int(st["step"]))
        L.check(L.lib().mipnerf_set_params(self.ctx.handle, self._params, ops._stream()), "mipnerf_set_params")

    def allreduce_stats(self):
        """(mean ms, count) of the event pairs recorded around the gradient all-reduce since the last call (synchronises)."""
        if not self._ar_events:
            return None, 0
        self._ar_events[-1][1].synchronize()
        ms = [a.elapsed_time(b) for a, b in self._ar_events]
        self._ar_events = []
        return sum(ms) / len(ms), len(ms)

    # ---- one optimisation step -----------------------------------------------------------------------------------------
    def __call__(self):
        mlp, st = self.model.mlp, self.opt.state[self.opt._key()]
        if not mlp.grads_are_flat():
            raise RuntimeError("GraphedTrainStep: the parameters / gradients are no longer views of the flat buffers")
        if self.use_graph and self._graphs is None:
            self._capture()          # may fall back to eager (sets use_graph False)
        if not self.collective:
            if self.use_graph:
                self._graphs[0].replay()
            else:
                self._fwd_bwd()
                self._update()
        else:
            if self.use_graph:
                self._graphs[0].replay()
            else:
                self._fwd_bwd()
            # one SUM all-reduce of the 2.45 MB flat gradient; c10d runs it on its own stream, ordered after the backward
            # by an event, so the host goes on (next batch's ray generation) while it is in flight; wait() orders the Adam
            # kernels behind it.  The mean (1 / world) is applied inside the Adam kernel.
            if self.time_allreduce and len(self._ar_events) < 4096:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                work = dist.all_reduce(mlp._flat_grad, op=dist.ReduceOp.SUM, async_op=True)
                work.wait()
                ev[1].record()             # on the launch stream, behind the all-reduce it now waits for
                self._ar_events.append(ev)
            else:
                work = dist.all_reduce(mlp._flat_grad, op=dist.ReduceOp.SUM, async_op=True)
                work.wait()
            if self.use_graph:
                self._graphs[1].replay()
            else:
                self._update()
        st["step"] += 1
        mlp._flat_grad_valid = False        # consumed: the next backward overwrites
        # master parameters changed behind torch's back (no version bump) and the streams were re-packed in the graph
        self.ctx._packed_key = tuple((p.data_ptr(), p._version) for p in mlp.ordered_params())
        return self.scalars
