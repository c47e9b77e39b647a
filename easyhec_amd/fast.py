"""``FusedPoseStep`` -- one optimisation step of /root/reference/easyhec/trainer/rbsolver.py:29-43 as a fixed chain of
three HIP launches with no host round trip: [pose forward + vertices + raster records] -> jobs (coverage, depth where the
silhouette analysis will look, and the resolve of every job by the wave that drew it) -> composite [+ in its last
workgroup: loss, gradients, pose backward, Adam]; data-parallel: the composite launch stops before Adam and ONE more
launch exchanges the 8 floats with the peers and applies Adam (peer-memory mailboxes; or an all-reduce followed by Adam).

It operates IN PLACE on an :class:`easyhec_amd.rb_solver.RBSolver`'s ``dof`` parameter and ``history_ops`` buffer and
keeps torch.optim.Adam-compatible state (exp_avg, exp_avg_sq, step), so it is interchangeable with the autograd path
of :class:`easyhec_amd.trainer.RBSolverTrainer` step for step (tests/test_gpu_fast.py)."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib, fused

__all__ = ["FusedPoseStep"]


def _f(x):
    return ctypes.c_float(float(x))


def ranks_agree(ok, pg, dev):
    """True iff EVERY rank of the process group passes ``ok`` = True (one all-reduce(min) of a flag; collective: every
    rank must call it)."""
    on_dev = dist.get_backend(pg) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if on_dev else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
    return bool(int(flag.item()) == 1)


class FusedPoseStep:
    def __init__(self, model, batch, lr=0.003, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0005, near=0.001, far=10.0,
                 process_group=None, rccl=None, slack=None, p2p=None):
        self.model = model
        self.renderer = model._ensure_renderer()
        self.scene = model._ensure_scene()
        self.glctx = self.renderer.glctx
        dev = model.dof.device
        self.dev = dev
        self.H, self.W = model.H, model.W
        # a private copy, always (``.to`` / ``.contiguous`` return the caller's own tensor when nothing has to change, and
        # the sums cached by ``bind_ref`` below must not go stale under an in-place edit of ``batch["mask"]``)
        self.ref = batch["mask"].to(dev, torch.float32).contiguous().clone()
        self.link_poses = batch["link_poses"].to(dev, torch.float32).contiguous()
        self.K = batch["K"][0].to(dev, torch.float32).contiguous()
        self.B, self.L = self.link_poses.shape[0], self.link_poses.shape[1]
        assert self.L == self.scene.num_links and self.ref.shape == (self.B, self.H, self.W)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.near, self.far = near, far
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1
        # How the ranks exchange the 8-float vector: "rccl" = ncclAllReduce on the library's own communicator, enqueued on
        # the chain's stream between the solver step and Adam (default for the nccl backend, i.e. one process per GPU);
        # otherwise torch.distributed.all_reduce (gloo: the CPU tests, two ranks sharing one GPU).  rccl=True without a
        # process group makes a single-rank communicator (the same launch sequence on one GPU).
        # (EHR_TRY_RCCL=1, test hook: attempt the library-owned exchange under any backend, so that the agreement / fall-back
        #  branches below run where RCCL cannot come up -- two ranks on one device)
        import os as _os
        self.rccl = bool(rccl) if rccl is not None else (
            self.distributed and (dist.get_backend(self.pg) == "nccl" or _os.environ.get("EHR_TRY_RCCL") == "1"))
        # "p2p" (EHR_COMM=p2p, or p2p=True): the one-shot exchange over peer memory (ehr_comm_p2p_*: every rank stores its 8
        # floats into every peer's mailbox, sums in rank order and runs Adam in the SAME kernel -- one launch instead of an
        # all-reduce plus ehr_pose_adam).  Needs the ranks on GPUs of one node (or, for tests, on one GPU); set up over the
        # process group, checked against it, and dropped by ALL ranks together if any rank cannot set it up.
        self.p2p = False
        # Default under the nccl backend (one process per GPU of a node): tried first, the RCCL all-reduce is the fall-back.
        # EHR_COMM = p2p / rccl / torch forces the choice (p2p: under any backend, e.g. two test ranks on one device).
        comm_env = _os.environ.get("EHR_COMM", "")
        want_p2p = p2p if p2p is not None else (comm_env == "p2p" or (comm_env == "" and rccl is None and self.distributed
                                                                       and dist.get_backend(self.pg) == "nccl"))
        if comm_env in ("torch", "rccl") and p2p is None:
            want_p2p = False
        if comm_env == "torch" and rccl is None:
            self.rccl = False
        if want_p2p and self.distributed:
            ok, why = True, ""
            try:
                self._init_p2p()
            except RuntimeError as e:
                ok, why = False, str(e)
            if not ranks_agree(ok, self.pg, self.dev):
                if ok:
                    why = "another rank could not set it up"
                    with torch.cuda.device(self.dev):
                        _lib.lib().ehr_comm_p2p_close(self.glctx.handle)
                if p2p:
                    raise RuntimeError(f"peer-memory exchange unavailable: {why}")
                import sys
                print(f"[easyhec_amd] peer-memory exchange unavailable ({why}); using the all-reduce", file=sys.stderr)
            else:
                self.p2p = True
                self.rccl = False
        if self.rccl:
            ok, why = True, ""
            try:
                self._init_comm()
            except RuntimeError as e:
                if rccl:  # asked for explicitly: fail loudly
                    raise
                ok, why = False, str(e)
            if self.distributed and rccl is None:
                # the ranks must agree on the exchange they use: one rank on torch.distributed and the others on the
                # library's communicator would wait for each other for ever
                if ok and not ranks_agree(ok, self.pg, self.dev):
                    ok, why = False, "another rank could not set it up"
                    with torch.cuda.device(self.dev):
                        _lib.lib().ehr_comm_destroy(self.glctx.handle)
                elif not ok:
                    ranks_agree(False, self.pg, self.dev)
            if not ok:
                import sys
                print(f"[easyhec_amd] library-owned RCCL exchange unavailable ({why}); using torch.distributed.all_reduce",
                      file=sys.stderr)
                self.rccl = False
        # optimiser state (torch.optim.Adam names): a fresh Adam (step 0, zero moments) unless load_state_dict restores
        # one -- like the reference's load_model path.  The row of ``history_ops`` the next step records its pose in is a
        # counter of its own (the reference's first all-zero row, rb_solver.py:50-51): it starts at the model's history
        # cursor, so a solver built on a loaded checkpoint appends whatever the optimiser's step count is.
        self.exp_avg = torch.zeros(6, device=dev)
        self.exp_avg_sq = torch.zeros(6, device=dev)
        self.step_t = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.hist_row = torch.full((1,), int(model.history_cursor()), dtype=torch.int32, device=dev)
        # work buffers, allocated once
        self.mvp = torch.empty((self.B, self.L, 4, 4), device=dev)
        self.grad_mvp = torch.empty((self.B, self.L, 4, 4), device=dev)
        self.tc_jac = torch.empty((7, 16), device=dev)
        self.loss_b = torch.empty((self.B,), device=dev)
        self.red = torch.empty((8,), device=dev)
        self.loss = torch.zeros((1,), device=dev)
        self.grad = torch.zeros((6,), device=dev)
        self.mask = torch.empty((self.B, self.H, self.W), device=dev)
        # job slots: `slack` per view tile (default: half as many slots as a view has tiles) instead of one per (view,
        # link, tile) -- 50 MB instead of 0.8 GB of scratch at 8 views 720p x 8 links; the workloads here use a tenth of
        # that (a robot's links touch ~5 % of a frame's tiles).  A view that needs more (every pixel under more than
        # `slack` link boxes on average: a close-up) is REPORTED -- NaN loss, dof and Adam state untouched -- and
        # :meth:`recover_from_overflow` plans again with a slot for every (view, link, tile).  EHR_VB_SLACK overrides (0 = all).
        # (Small images: at least 256 slots per view -- a link's box touches a few tiles however small the frame is.)
        import os
        if slack is None and "EHR_VB_SLACK" not in os.environ:
            ntiles = ((self.W + 31) // 32) * ((self.H + 7) // 8)
            self.slack = max(0.5, 256.0 / ntiles)
        else:
            self.slack = float(os.environ["EHR_VB_SLACK"]) if slack is None else float(slack)
        fused._ensure_plan(self.glctx, self.scene, self.B, self.H, self.W, slack=self.slack)
        # the reference masks are constants of the solve: cache the loss of the tiles no link touches once
        # (ehr_fused_bind_ref; bit-identical results).  self.ref is this object's private copy, never written to.
        fused.bind_ref(self.glctx, self.scene, self.ref)
        self._graph = None
        # A step the chain REPORTS (NaN loss: slot-limited plan overflowed / the view needs the general-triangle pass) leaves
        # dof and Adam untouched, so nothing is lost but time -- unless nobody looks.  step() therefore looks itself, without
        # ever waiting: every `check_every` steps the loss goes to pinned host memory behind an event, the copy that was
        # started `check_every` steps earlier is inspected, and a NaN there triggers recover_from_overflow().  Callers that
        # need an exact number of EFFECTIVE steps ask steps_done (Adam's own counter: it only advances on real steps).
        self.check_every = 16
        self._calls = 0
        self._probe = torch.zeros(1, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(1)
        self._probe_ev = None
        self.recoveries = []  # what recover_from_overflow() did, in order
        self._calls0, self._steps0 = 0, 0  # step() calls / Adam steps at the last point both were known (mark_counts)

    def _init_comm(self):
        """ncclCommInitRank through the C ABI (``ehr_comm_*``): rank 0's ncclUniqueId travels over the process group that
        is already up (the only use torch.distributed has on this path)."""
        lib = _lib.lib()
        world = dist.get_world_size(self.pg) if self.distributed else 1
        rank = dist.get_rank(self.pg) if self.distributed else 0
        idbuf = (ctypes.c_ubyte * 128)()
        err = None
        if rank == 0:
            try:
                _lib.check(lib.ehr_comm_unique_id(idbuf), "ehr_comm_unique_id")
            except RuntimeError as e:   # the other ranks are waiting in the broadcast below: they get an all-zero id
                err, idbuf = e, (ctypes.c_ubyte * 128)()
        if world > 1:
            on_dev = dist.get_backend(self.pg) == "nccl"
            t = torch.tensor(list(idbuf), dtype=torch.uint8, device=self.dev if on_dev else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            idbuf = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        if err is not None or not any(idbuf):
            raise RuntimeError(f"no ncclUniqueId from rank 0 ({err})")
        with torch.cuda.device(self.dev):
            _lib.check(lib.ehr_comm_init(self.glctx.handle, idbuf, world, rank), "ehr_comm_init")
            # self-check before the solve depends on it: one all-reduce of a known vector on the new communicator must
            # give what the process group gives (the N > 1 form of this exchange has never run on hardware in the build
            # container; a mismatch here raises, and the default selection above falls back to torch.distributed)
            probe = torch.full((8,), float(rank + 1), device=self.dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(lib.ehr_comm_allreduce(self.glctx.handle, _lib.ptr(probe), 8, stream), "ehr_comm_allreduce")
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0
            if not bool((probe == want).all()):
                raise RuntimeError(f"ehr_comm_allreduce self-check: got {probe.tolist()}, expected {want}")

    def _init_p2p(self):
        """Mailboxes of the one-shot exchange (``ehr_comm_p2p_*``): every rank exports its mailbox's 64-byte IPC handle, the
        handles travel over the process group that is already up, every rank opens its peers', and one exchange of a known
        vector must give what it should before the solve depends on it.  Every rank makes the same collective calls whatever
        fails locally (a rank that cannot export ships an all-zero handle; the open phase ends on an agreement)."""
        lib = _lib.lib()
        world, rank = dist.get_world_size(self.pg), dist.get_rank(self.pg)
        hbuf, err = (ctypes.c_ubyte * 64)(), None
        try:
            with torch.cuda.device(self.dev):
                _lib.check(lib.ehr_comm_p2p_export(self.glctx.handle, hbuf), "ehr_comm_p2p_export")
        except RuntimeError as e:
            hbuf, err = (ctypes.c_ubyte * 64)(), e
        on_dev = dist.get_backend(self.pg) == "nccl"
        mine = torch.tensor(list(hbuf), dtype=torch.uint8, device=self.dev if on_dev else "cpu")
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=self.pg)          # (also the barrier "every rank has exported")
        flat = torch.stack(every).cpu().contiguous()
        if err is None and not bool((flat != 0).any(dim=1).all()):
            err = RuntimeError("a rank exported no mailbox")
        if err is None:
            try:
                allh = (ctypes.c_ubyte * (64 * world))(*flat.view(-1).tolist())
                with torch.cuda.device(self.dev):
                    _lib.check(lib.ehr_comm_p2p_open(self.glctx.handle, allh, world, rank), "ehr_comm_p2p_open")
            except RuntimeError as e:
                err = e
        if not ranks_agree(err is None, self.pg, self.dev):   # every rank has opened (or nobody goes on): stores may begin
            raise err if err is not None else RuntimeError("another rank could not open the mailboxes")
        with torch.cuda.device(self.dev):
            probe = torch.full((8,), float(rank + 1), device=self.dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(lib.ehr_comm_p2p_step(self.glctx.handle, _lib.ptr(probe), None, None, None, None, _f(0), _f(0), _f(0),
                                             _f(0), _f(0), None, None, stream), "ehr_comm_p2p_step")
            torch.cuda.synchronize()
        want = world * (world + 1) / 2.0
        if not bool((probe == want).all()):
            raise RuntimeError(f"ehr_comm_p2p_step self-check: got {probe.tolist()}, expected {want}")

    # -- one step -------------------------------------------------------------------------------------------------
    def _enqueue(self, want_mask, stream=None):
        lib = _lib.lib()
        if stream is None:
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        m, sc = self.model, self.scene
        dof = m.dof.data
        hist = m.history_ops
        # one C call = 3 launches: [pose fwd + vertices + raster records] -> jobs, resolved by the waves that drew them
        # [-> general-triangle jobs + their resolve, once a step has needed them] -> composite [+ in its last workgroup:
        # accumulators + pose bwd (+ Adam)]
        _lib.check(lib.ehr_solver_step(
            self.glctx.handle, _lib.ptr(sc.verts), _lib.ptr(sc.tris), _lib.ptr(sc.tri_link), _lib.ptr(sc.vert_link),
            _lib.ptr(sc.opp), _lib.ptr(self.K), _lib.ptr(self.link_poses), _lib.ptr(self.ref), self.B, self.L,
            sc.num_verts, sc.num_tris, self.H, self.W, _f(self.near), _f(self.far), _lib.ptr(dof),
            _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), _lib.ptr(self.step_t), _lib.ptr(hist), hist.shape[0],
            _lib.ptr(self.hist_row), _f(self.lr), _f(self.betas[0]), _f(self.betas[1]), _f(self.eps), _f(self.wd), _lib.ptr(self.mvp),
            _lib.ptr(self.tc_jac), _lib.ptr(self.mask if want_mask else None), _lib.ptr(self.loss_b),
            _lib.ptr(self.grad_mvp), _lib.ptr(self.red), _lib.ptr(self.loss), _lib.ptr(self.grad),
            int(self.distributed or self.rccl), stream), "ehr_solver_step")
        if self.p2p:
            # the exchange and Adam in ONE launch: stores into the peers' mailboxes, a wait on the own one, sums in rank order
            _lib.check(lib.ehr_comm_p2p_step(self.glctx.handle, _lib.ptr(self.red), _lib.ptr(dof), _lib.ptr(self.exp_avg),
                                             _lib.ptr(self.exp_avg_sq), _lib.ptr(self.step_t), _f(self.lr), _f(self.betas[0]),
                                             _f(self.betas[1]), _f(self.eps), _f(self.wd), _lib.ptr(self.loss),
                                             _lib.ptr(self.grad), stream), "ehr_comm_p2p_step")
        elif self.distributed or self.rccl:
            # the ONE collective of a step (32 bytes), between the chain and Adam
            if self.rccl:
                _lib.check(lib.ehr_comm_allreduce(self.glctx.handle, _lib.ptr(self.red), 8, stream), "ehr_comm_allreduce")
            else:
                dist.all_reduce(self.red, op=dist.ReduceOp.SUM, group=self.pg)
            _lib.check(lib.ehr_pose_adam(_lib.ptr(dof), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                         _lib.ptr(self.step_t), _lib.ptr(self.red), _f(self.lr), _f(self.betas[0]),
                                         _f(self.betas[1]), _f(self.eps), _f(self.wd), _lib.ptr(self.loss),
                                         _lib.ptr(self.grad), stream), "ehr_pose_adam")

    def step(self, want_mask=False):
        """Enqueue one optimisation step.  Returns the (device, 1-element) mean mask loss evaluated BEFORE the update,
        like ``loss`` in trainer/rbsolver.py:33-41.  Never synchronises."""
        self.model._hist_n = None  # the chain writes history_ops rows itself: the host cursor is stale from here on
        with torch.cuda.device(self.dev):
            if self._graph and not want_mask:
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(_lib.lib().ehr_graph_launch(self.glctx.handle, stream), "ehr_graph_launch")
            else:
                self._enqueue(want_mask)
            self._calls += 1
            if self._calls % self.check_every == 0:
                self._poll()
        return self.loss

    def mark_counts(self):
        """Remember how many step() calls and how many Adam steps there have been (synchronises): what recover_from_overflow
        measures the reported -- i.e. not taken -- steps against.  Called wherever the Adam counter is set from outside."""
        self._calls0, self._steps0 = self._calls, int(self.step_t.item())

    def _count_reported(self):
        """How many step() calls since ``mark_counts`` were REPORTED steps (NaN loss: pose, Adam moments and counter
        untouched).  Their ``history_ops`` rows need no rewinding: the chain's head notices that Adam's counter has not moved
        since the previous step and writes the (unchanged) pose to the SAME row again, so the history holds one row per
        effective step on every rank, however late a rank looks (include/ehr.h, ehr_solver_step)."""
        lost = (self._calls - self._calls0) - (int(self.step_t.item()) - self._steps0)
        self.model._hist_n = None
        self.mark_counts()
        return max(lost, 0)

    def _poll(self):
        """Non-blocking look at the loss of the step taken `check_every` steps ago; starts the next look."""
        if self._probe_ev is not None:
            if not self._probe_ev.query():
                return  # (still in flight: look again next time; never wait here)
            self._probe_ev = None
            v = float(self._probe[0])
            if v != v:
                what = self.recover_from_overflow()
                if what:
                    self.recoveries.append(what)
        self._probe.copy_(self.loss, non_blocking=True)
        self._probe_ev = torch.cuda.Event()
        self._probe_ev.record()

    def capture(self):
        """Record the step's launch chain (3 kernels on one stream; 4-5 data-parallel) into a hipGraph owned by the rasterizer
        context (``ehr_graph_*`` in include/ehr.h); ``step()`` then replays it with one host call.  Iteration state lives on the device, so replays are ordinary optimisation steps.  The
        chain is GPU-bound, so this saves host time, not step time.  The data-parallel step is captured too when its
        exchange is the library's own -- the peer-memory one (``p2p``: [solver step, exchange + Adam]) or ncclAllReduce
        (``rccl``: [solver step, all-reduce, Adam]) -- on one stream; with the torch.distributed exchange (gloo) it cannot be."""
        if self._graph:
            return
        if self.distributed and not (self.rccl or self.p2p):
            raise RuntimeError("capture(): not available with the torch.distributed exchange (use the RCCL or the peer-memory one)")
        lib = _lib.lib()
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize()
            cap = ctypes.c_void_p()
            _lib.check(lib.ehr_graph_begin(self.glctx.handle, ctypes.byref(cap)), "ehr_graph_begin")
            try:
                self._enqueue(False, stream=cap)
            except Exception:
                lib.ehr_graph_release(self.glctx.handle)
                raise
            _lib.check(lib.ehr_graph_end(self.glctx.handle), "ehr_graph_end")
        self._graph = True

    def recover_from_overflow(self):
        """Call when a step's loss came back NaN.  Synchronises.  If the context reports that the step needed the
        general-triangle pass (EHR_ERR_RETRY: the context launches it from now on) the graph, if any, is re-captured; if it
        reports an overflow and the plan was slot-limited, plans again with a slot for every (view, link, tile), re-binds
        the reference masks and re-captures.  Returns what it did (a non-empty string) in both cases, False if the context
        reports nothing: the steps since the overflow changed nothing (dof, Adam moments and step
        counter stay untouched on a NaN), so the caller simply goes on stepping.  Raises on any other overflow."""
        with torch.cuda.device(self.glctx.device):
            rc = _lib.lib().ehr_fused_status(self.glctx.handle)
        # (on every rank of a data-parallel job alike, whichever rank's views caused the report: the reduced loss was NaN for
        #  all of them, none of them stepped, and each of them kept recording the unchanged pose in one and the same row)
        self._count_reported()
        if rc == 0:
            return False
        had_graph = bool(self._graph)
        if rc == _lib.EHR_ERR_RETRY:
            # the step met triangles for the general-triangle pass, which the chain had not been launching: the context
            # has switched it on; a captured chain is recorded again with it
            if had_graph:
                self.release_graph()
                self.capture()
            return "general-triangle pass"
        if self.slack == 0.0:
            _lib.check(rc, "fused render")
        self.release_graph()
        self.slack = 0.0
        fused._ensure_plan(self.glctx, self.scene, self.B, self.H, self.W, slack=0.0)
        fused.bind_ref(self.glctx, self.scene, self.ref)
        if had_graph:
            self.capture()
        return "job slots"

    def release_graph(self):
        if self._graph:
            _lib.check(_lib.lib().ehr_graph_release(self.glctx.handle), "ehr_graph_release")
            self._graph = None

    @property
    def steps_done(self):
        return int(self.step_t.item())

    def state_dict(self):
        """torch.optim.Adam-shaped state for checkpoints (trainer/rbsolver.py:95-114)."""
        return {"state": {0: {"step": self.step_t.float().cpu().reshape(()), "exp_avg": self.exp_avg.cpu().clone(),
                              "exp_avg_sq": self.exp_avg_sq.cpu().clone()}},
                "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd,
                                  "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                                  "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                                  "params": [0]}]}

    def load_state_dict(self, sd):
        """Inverse of :meth:`state_dict`; also accepts a ``torch.optim.Adam.state_dict()`` of the same parameter."""
        st = sd.get("state", {})
        if len(st) == 0:
            return  # a fresh optimiser
        st = st[sorted(st.keys())[0]]
        self.exp_avg.copy_(torch.as_tensor(st["exp_avg"], dtype=torch.float32).reshape(6))
        self.exp_avg_sq.copy_(torch.as_tensor(st["exp_avg_sq"], dtype=torch.float32).reshape(6))
        self.step_t.fill_(int(round(float(torch.as_tensor(st["step"]).reshape(-1)[0]))))
        self.mark_counts()
