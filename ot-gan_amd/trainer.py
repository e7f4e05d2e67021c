"""The OT-GAN training step (reference train.py:52-151 graph section + :207-226 hot loop),
wired on the HIP kernels: generator / critic forward, mini-batch Sinkhorn matching, gradient
injection (`grad_ys`), cross-rank gradient sum, optimiser and EMA updates."""
import importlib
import os

import torch

from . import parallel
from .utils import matching, nn


def _lib_prof_on():
    from . import _lib
    return _lib.prof_enabled()


class _frozen:
    """Context manager: the given leaf tensors do not require grad inside the block."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)
        return False


class _Region:
    """`with model._timed(name):` -- two events on the current stream around a region of the step, only while timers are
    enabled (bench.py, ranks > 1); no synchronisation until `collect_timers()`."""
    __slots__ = ("pairs", "e0")

    def __init__(self, pairs):
        self.pairs = pairs

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.pairs.append((self.e0, e1))
        return False


class _NoRegion:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_REGION = _NoRegion()


class OTGAN:
    """State of one training run.  `args` carries the reference's flags (train.py:14-33)
    plus: image_size, matching_scope ('global' = one OT problem set over all ranks, the
    reference's semantics; 'local' = an independent problem set per rank)."""

    def __init__(self, args, device, init_batch=None):
        """init_batch: [n, H, W, 3] data for the reference's intended data-dependent initialisation pass
        (only with args.data_dependent_init; nn.py:133-162, train.py:52-54)."""
        self.args = args
        self.device = device
        self.rank, self.world = parallel.get_rank(), parallel.world_size()
        if args.nr_gpu % 2 != 0:
            raise AssertionError("nr_gpu must be even (train.py:34)")
        if args.nr_gpu % self.world != 0:
            raise ValueError("--nr_gpu (logical shards) must be a multiple of the number of ranks")
        self.shards = args.nr_gpu // self.world           # logical shards on this rank
        self.scope = getattr(args, "matching_scope", "global")
        if self.world == 1:
            self.scope = "local"
        if self.scope == "global" and self.world % 2 != 0:
            # a rank's rows must lie inside ONE mini-batch half (shards [0,S/2) | [S/2,S), matching.py:16-19)
            raise ValueError(f"global matching scope needs an even number of ranks (got {self.world}): with an odd "
                             "count one rank's shards straddle the two mini-batch halves; use --matching_scope local")
        # world size 1 with OTGAN_FORCE_COLLECTIVES=1: run the RCCL calls anyway (single-GPU readiness test)
        self.collectives = not parallel._skip_collectives()
        if self.scope == "local" and self.shards % 2 != 0:
            raise ValueError("local matching needs an even number of shards per rank")
        self.nb = self.shards * args.batch_size             # images per rank per step
        mod = importlib.import_module(f".models.{args.model}", package=__package__)  # train.py:38-41
        self.generator, self.discriminator = mod.generator, mod.discriminator
        self.generator.reset(seed=args.seed, device=device)
        self.discriminator.reset(seed=args.seed, device=device)
        self.model_opts = {"nonlinearity": args.nonlinearity}
        size = getattr(args, "image_size", 32)
        if size != 32:
            if args.model != "dcgan":
                raise ValueError("--image_size other than 32 is only available for --model dcgan")
            self.model_opts["image_size"] = size
        # parameter creation pass (train.py:52-56; the data-dependent init it builds is never run by the reference:
        # default g = 1, b = 0; --data_dependent_init executes it on `init_batch`)
        ddi = bool(getattr(args, "data_dependent_init", False))
        if ddi and init_batch is None:
            raise ValueError("--data_dependent_init needs an initial data batch (OTGAN(args, device, init_batch=...))")
        nn.data_dependent_init(ddi)
        try:
            with torch.no_grad():
                x0 = init_batch.to(device) if ddi else torch.zeros(2, size, size, 3, device=device)
                f = self.discriminator(x0, init=True, **self.model_opts)
                self.generator(batch_size=x0.shape[0], init=True, device=device, **self.model_opts)
        finally:
            nn.data_dependent_init(False)
        self.num_features = f.shape[-1]
        # one flat buffer per network: optimiser / EMA / gradient all-reduce act on it in one go
        groups = (self.discriminator.flatten(), self.generator.flatten())
        if ddi and self.world > 1:
            # the data-dependent pass draws the generator's latent from the per-rank RNG stream (train.py seeds
            # seed + rank), so every rank computed its own g / b statistics: the replicas would start different and
            # never meet (only gradients are exchanged).  Rank 0's initialisation is the run's.
            from . import ops
            for grp in groups:
                parallel.broadcast_(grp.flat, src=0)
            ops.bump_weights_epoch()
        self.disc_params = self.discriminator.trainable_variables()     # train.py:61
        self.gen_params = self.generator.trainable_variables()          # train.py:62
        self.ema = nn.ExponentialMovingAverage(decay=0.999)             # train.py:63
        self.maintain_averages = self.ema.apply(self.gen_params)        # train.py:64
        mk = {"adam": nn.adam_updates, "adamax": nn.adamax_updates, "nesterov": nn.nesterov_updates}
        if args.optimizer not in mk:
            raise ValueError("unsupported optimizer")
        kw = {"mom1": 0.5} if args.optimizer == "nesterov" else {"mom1": 0.5, "mom2": 0.999}
        # Exchange schedule (parallel.collectives_mode; round 6: decided at start-up by a co-residency self-check on every rank
        # -- RCCL collectives beside the 256 x 128 GEMM, bit-compared -- unless OTGAN_COLLECTIVES pins it).  "overlapped": the
        # gradient all-reduce leaves bucket by bucket underneath the backward pass (parallel.GradBuckets) and the real
        # features' all-gather starts right behind the critic's pass over the real batch; "serial": one flat all-reduce after
        # the backward pass, every gather waited for where it is issued.  The second stream (below) runs in both.
        if self.collectives:
            parallel.resolve_collectives_mode(device)
        self.collectives_mode = parallel.collectives_mode() if self.collectives else "none"
        side_ok = os.environ.get("OTGAN_SIDE_STREAM", "1") != "0"
        overlap = self.collectives and self.collectives_mode == "overlapped"
        self.disc_buckets = parallel.GradBuckets(self.disc_params) if overlap else None
        self.gen_buckets = parallel.GradBuckets(self.gen_params) if overlap else None
        self.gen_optimizer = mk[args.optimizer](self.gen_params, **kw)          # train.py:142
        self.disc_optimizer = mk[args.optimizer](self.disc_params, **kw)        # train.py:143
        # the generator's Adam step can carry the EMA of the updated weights along (one launch, one read of p less)
        self.ema_fused = self.gen_optimizer.fuse_ema(self.ema)
        self.step_counter = 0
        self.last = {}
        import collections
        self._max_ahead = 2          # steps the host may run ahead of the device (_throttle; 0 = unbounded)
        self._step_ends, self._free_events = collections.deque(), []
        self.timers = None        # name -> [(start event, end event)]; see enable_timers()
        self._d_weights_in_graph = False
        # Two chains of every step are independent of its critical path and run on a SECOND STREAM (round 5; single-process
        # runs; OTGAN_SIDE_STREAM=0 = everything on one stream): the critic's pass over the real batch of a generator step
        # (under the generator's forward pass) and every layer's weight-gradient chain (under the input-gradient chain of the
        # layers in front of it).  Same kernels, same arguments, bit-identical results (tests/test_side_stream_gpu.py); the
        # chains alternate HBM-bound transforms and matrix-bound GEMMs and, on one in-order stream, every kernel also waits
        # for its predecessor's last workgroup: A/B/A/B on one box 8.95 / 8.96 -> 8.48 / 8.50 ms per DCGAN step, DenseNet
        # 29.2 -> 27.4 ms (profiles/r05_side_stream_ab.txt).  Captured with the step under step graphs.  With overlapped collectives the bucket
        # hooks copy and all-reduce on the side stream (parallel.GradBuckets._on_grad) and the real features' all-gather is
        # issued from it (below): the main stream never waits for the side stream inside a pass.
        on = side_ok
        self.fork_real_pass = self.fork_wgrad = on
        self._side_stream = torch.cuda.Stream(device=device) if on else None
        # Whole steps as hipGraphs (GraphedSteps).  Default since round 6: ON for --model densenet (a launch-bound step: ~600
        # launches, replay 25.5 ms against 27.6 - 29.6 ms eager), OFF for dcgan (replay 8.60 ms against 8.52 - 8.56 ms eager);
        # --step_graph / OTGAN_STEP_GRAPH=1 force it on, OTGAN_STEP_GRAPH=0 off.  Single-process runs only (gloo cannot be
        # captured; RCCL under capture is untested here).  The second stream's chains are captured with the step.
        env = os.environ.get("OTGAN_STEP_GRAPH")
        asked = getattr(args, "step_graph", None)
        want = (env == "1") if env in ("0", "1") else (args.model == "densenet" if asked is None else bool(asked))
        self.graphs = GraphedSteps(self) if (want and not self.collectives and self.world == 1) else None

    # ---------------------------------------------------------------- per-region step times (bench.py, ranks > 1)
    def enable_timers(self, on=True):
        """Event-time the exchange steps of the data-parallel step on this rank's compute stream: `allgather` (feature
        all-gathers and the cost-slice all-gather, including what the stream waits for them), `matching` (the rank's cost
        row slices, the Sinkhorn problems and the plans applied to its rows: `_match` without its gathers) and `allreduce`
        (the gradient SUM).  In the default serial schedule the stream waits for every collective where it is issued,
        so the regions do not overlap compute and their sum is the non-scaling part of the step."""
        self.timers = {} if on else None
        if on and self.graphs is not None:
            self.graphs.drop("region timers need the eager step")
            self.graphs = None

    def _timed(self, name):
        if self.timers is None:
            return _NO_REGION
        return _Region(self.timers.setdefault(name, []))

    def collect_timers(self, steps):
        """-> {name_ms: mean milliseconds per step over `steps` steps}; synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in (self.timers or {}).items():
            out[name + "_ms"] = round(sum(a.elapsed_time(b) for a, b in pairs) / max(1, steps), 4)
        if "match_total_ms" in out:
            out["matching_ms"] = round(out.pop("match_total_ms") - out.get("allgather_in_match_ms", 0.0), 4)
        out["allgather_ms"] = round(out.pop("allgather_in_match_ms", 0.0) + out.pop("allgather_early_ms", 0.0), 4)
        self.timers = {} if self.timers is not None else None
        return out

    def sinkhorn_rows(self):
        """Rows N of one Sinkhorn problem of this run (matching.py:16-19: the shards of the matching scope form two
        halves; --single_batch solves over all of them, matching.py:91-93)."""
        S = self.args.nr_gpu if (self.scope == "global" and self.world > 1) else self.shards
        return S * self.args.batch_size if self.args.single_batch else (S // 2) * self.args.batch_size

    # ---------------------------------------------------------------- matching (train.py:88-98)
    def _match(self, f_gen, f_dat, pending_dat=None, need_dat=True):
        """-> (grad_gen, grad_dat or None, distance, entropy): the injected upstream gradients of train.py:111,125-126
        (un-normalised matched differences) for this rank's samples.  `need_dat=False` (generator steps) skips the
        data-side gradient."""
        a = self.args
        plain = not (a.single_batch or a.no_sinkhorn)
        if self.scope == "global" and self.world > 1:
            S = self.world * self.shards
            with self._timed("allgather_in_match"):
                allg = parallel.all_gather_rows(f_gen)
                alld = pending_dat.wait() if pending_dat is not None else parallel.all_gather_rows(f_dat)
            if plain:
                # The cost matrices are row-sharded like the reference (matching.py:29-39): a rank
                # of the first half computes its rows of (a1,a2) (a1,b1) (a1,b2), a rank of the
                # second half its rows of (b2,b1) (a2,b1) (a2,b2); the slices are all-gathered.
                # Every rank then solves the six (small, on-chip) Sinkhorn problems and applies the
                # plans only to the rows of its own samples.
                if rank_stack_ok(self.nb, allg):
                    # round 5: the gathered features are split into the GEMM engine's operand ONCE for both library calls
                    g_gen, g_dat, ent, dist = rank_matching_stack(self.rank, self.world, self.nb, allg, alld, a.sinkhorn_lambda,
                                                                  a.nr_sinkhorn_iter, need_dat, self._gather_slices)
                    return g_gen, g_dat, dist, ent
                K = self._sharded_log_kernels(f_gen, f_dat, allg, alld)
                g_gen, g_dat, ent, dist = matching.matched_feature_grads(
                    allg, alld, a.sinkhorn_lambda, a.nr_sinkhorn_iter, need_b=need_dat,
                    rows=(self.rank * self.nb, self.nb), log_kernels=K)
                return g_gen, g_dat, dist, ent
            if a.single_batch:
                # --single_batch in the global scope, row-sharded like the reference (matching.py:99-104): this rank's rows
                # of the a-a, b-b and a-b costs, all-gathered; every rank solves the three problems and applies the plans to
                # its own rows only (training-mode entry: the injected differences directly)
                K = self._sharded_single_log_kernels(f_gen, f_dat, allg, alld)
                g_gen, g_dat, ent, dist = matching.matched_feature_grads_single_batch(
                    allg, alld, a.sinkhorn_lambda, a.nr_sinkhorn_iter, need_b=need_dat,
                    rows=(self.rank * self.nb, self.nb), log_kernels=K)
                return g_gen, g_dat, dist, ent
            fa, fb = list(torch.chunk(allg, S, 0)), list(torch.chunk(alld, S, 0))
        else:
            if pending_dat is not None:       # forced-collective mode at world size 1: same rows, via RCCL
                f_dat = pending_dat.wait()
            if plain:
                g_gen, g_dat, ent, dist = matching.matched_feature_grads(
                    f_gen, f_dat, a.sinkhorn_lambda, a.nr_sinkhorn_iter, need_b=need_dat)
                return g_gen, g_dat, dist, ent
            if a.single_batch:
                g_gen, g_dat, ent, dist = matching.matched_feature_grads_single_batch(
                    f_gen, f_dat, a.sinkhorn_lambda, a.nr_sinkhorn_iter, need_b=need_dat)
                return g_gen, g_dat, dist, ent
            fa = list(torch.chunk(f_gen, self.shards, 0))
            fb = list(torch.chunk(f_dat, self.shards, 0))
        if a.single_batch:      # (not reached: both scopes take the training-mode entry above)
            m = matching.get_matched_features_single_batch(fa, fb, a.sinkhorn_lambda, a.nr_sinkhorn_iter)
        else:
            m = matching.get_matched_features_random(fa, fb)
        dist = m.distance if m.distance is not None else matching.calc_distance(fa, fb, m)
        lo = self.rank * self.shards if (self.scope == "global" and self.world > 1) else 0
        pick = lambda lst: torch.cat(lst[lo:lo + self.shards], 0)
        # injected upstream gradients (train.py:111,125-126): un-normalised matched differences
        grad_gen = pick(m[0]) - pick(m[2])
        grad_dat = pick(m[1]) - pick(m[3])
        return grad_gen, grad_dat, dist, m[4]

    def _gather_slices(self, mine):
        with self._timed("allgather_in_match"):
            return parallel.all_gather_rows(mine.unsqueeze(0))                          # [W,3,nb,N]

    def _sharded_log_kernels(self, f_gen, f_dat, fa, fb):
        mine = rank_log_kernel_slices(self.rank, self.world, f_gen, f_dat, fa, fb, self.args.sinkhorn_lambda)
        with self._timed("allgather_in_match"):
            allk = parallel.all_gather_rows(mine.unsqueeze(0))                          # [W,3,nb,N]
        return assemble_log_kernels(allk, self.world)

    def _sharded_single_log_kernels(self, f_gen, f_dat, allg, alld):
        mine = rank_single_log_kernel_slices(f_gen, f_dat, allg, alld, self.args.sinkhorn_lambda)
        with self._timed("allgather_in_match"):
            allk = parallel.all_gather_rows(mine.unsqueeze(0))                          # [W,3,nb,n]
        return assemble_single_log_kernels(allk, self.args.sinkhorn_lambda)

    # ---------------------------------------------------------------- one sess.run
    def step(self, x_data, noise=None, apply_updates=True):
        """x_data: [shards*batch_size, H, W, 3] in [-1, 1].  Runs a critic step when
        step_counter % (nr_gen_per_disc+1) == 0, else a generator step (train.py:214-226).
        `noise` (tests) replaces the generator's own latent draw; `apply_updates=False`
        (tests) leaves the parameters untouched and returns the summed gradients.

        With step graphs (single-process runs; default for densenet) the step is captured in a hipGraph per step kind after one
        eager period and replayed (GraphedSteps below): the same launches with the same arguments."""
        assert x_data.shape[0] == self.nb
        period = self.args.nr_gen_per_disc + 1
        phase = self.step_counter % period
        kind = "disc" if phase == 0 else "gen"
        self._throttle()
        if self.graphs is not None and noise is None and apply_updates:
            done = self.graphs.run(x_data, phase)
            if done is not None:
                self.last = done
                self._mark_step_end()
                return self.last
        from . import ops
        ops.SIDE_STREAM = self._side_stream if self.fork_wgrad else None
        try:
            dist, ent, grads = self._step_body(x_data, kind, noise, apply_updates)
        finally:
            ops.SIDE_STREAM = None
        self._mark_step_end()
        if kind == "disc":
            self._d_weights_in_graph = False      # the critic changed: graphs that read its cached operands wait for a refresh
        self.step_counter += 1
        self.last = {"kind": kind, "distance": dist, "entropy": ent}      # device scalars, no sync
        if not apply_updates:
            self.last["grads"] = grads
        return self.last

    # The host never gets more than `_max_ahead` steps ahead of the device (default 2; 0: unbounded).  With
    # two streams a tensor that the other stream has used (record_stream) returns to torch's caching allocator only once
    # that stream's work on it has COMPLETED; a host that enqueues step after step never sees those completions and the
    # allocator answers with fresh hipMallocs -- four per step for the first ~150 steps of a run (bench.py: 120 device
    # allocations inside a 30-step window, the first critic step of the window 0.4 ms slow and once in ten runs 13 ms).
    # Waiting for the END of the step before the previous one costs nothing (a whole step is still queued behind it) and
    # puts the allocator in a steady state after two periods.
    def _throttle(self):
        if self._max_ahead > 0 and len(self._step_ends) >= self._max_ahead:
            ev = self._step_ends.popleft()
            ev.synchronize()
            self._free_events.append(ev)

    def _mark_step_end(self):
        if self._max_ahead > 0:
            ev = self._free_events.pop() if self._free_events else torch.cuda.Event()
            ev.record()
            self._step_ends.append(ev)
            if len(self._step_ends) > self._max_ahead:          # (callers that bypass step(): keep the queue bounded)
                self._free_events.append(self._step_ends.popleft())

    def _optimise(self, opt, grads, lr, critic):
        """The optimiser step (and the generator's EMA).  (Round 5 also tried it on a third stream, so that the next step's
        critic pass over a resident real batch starts under this HBM-bound launch: 8.46 / 8.53 ms without against 8.59 / 8.57 ms
        with -- dropped.)"""
        opt(grads, lr=lr)
        if not critic and not self.ema_fused:
            self.maintain_averages()

    def _backward(self, outputs, params, grad_outputs):
        """torch.autograd.grad with the layers' weight-gradient chains on the side stream (ops.SIDE_STREAM) when enabled."""
        from . import ops
        grads = torch.autograd.grad(outputs, params, grad_outputs)
        if self.fork_wgrad:
            ops.join_side_stream(grads)
        return grads

    def prepare_step_graphs(self, x_data):
        """Run training steps on `x_data` until every step kind is captured (one eager period, then each kind the first time
        it is due: at most 3 periods) -- the one-off setup of the replayed steps, for callers that time a window afterwards
        (bench.py).  Returns the number of steps run (0 when graphs are off)."""
        if self.graphs is None:
            return 0
        period = self.args.nr_gen_per_disc + 1
        kinds = {self.graphs._kind(p) for p in range(period)}
        n = 0
        while self.graphs is not None and self.graphs.dead is None and set(self.graphs.graphs) != kinds and n < 3 * period + 2:
            self.step(x_data)
            n += 1
        return n

    def _step_body(self, x_data, kind, noise=None, apply_updates=True):
        """The launches of one step (no host-side bookkeeping): -> (distance, entropy, summed gradients)."""
        a = self.args
        gkw = dict(self.model_opts)
        if noise is not None:
            gkw["noise"] = noise
        if kind == "disc":
            with torch.no_grad():
                ema = self.ema if a.train_disc_against_ema else None    # train.py:119-123
                x_gen = self.generator(batch_size=self.nb, ema=ema, device=self.device, **gkw)
            f_all = self.discriminator(torch.cat([x_data, x_gen], 0), **self.model_opts)
            f_dat, f_gen = f_all[:self.nb], f_all[self.nb:]
            with self._timed("match_total"):
                g_gen, g_dat, dist, ent = self._match(f_gen.detach(), f_dat.detach())
            if self.disc_buckets is not None:
                self.disc_buckets.arm()
            grads = self._backward(f_all, self.disc_params, torch.cat([g_dat, g_gen], 0))        # train.py:127-128
            with self._timed("allreduce"):
                grads = (self.disc_buckets.finish() if self.disc_buckets is not None
                         else parallel.allreduce_sum_(list(grads)))                               # train.py:134-139
            if apply_updates:
                self._optimise(self.disc_optimizer, grads, -a.learning_rate_disc, critic=True)    # train.py:143
        else:
            side = self._side_stream if self.fork_real_pass else None
            if side is not None:
                # the critic's pass over the real batch has no dependence on the generator: it runs on the second stream under
                # the generator's forward pass, so that each chain's kernels start in the other's tails and under-filled GEMM
                # grids (the 4x4 / 8x8 stages launch half a round of tiles) share the device
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side), torch.no_grad():
                    f_dat = self.discriminator(x_data, **self.model_opts)
                    # overlapped schedule: the real features are final here -- their all-gather starts now, ordered behind
                    # the side stream, and runs under the generator's forward pass and the second critic pass
                    pending = None
                    if self.collectives_mode == "overlapped" and (self.scope == "global" or self.world == 1):
                        with self._timed("allgather_early"):
                            pending = parallel.all_gather_rows_async(f_dat)
                f_dat.record_stream(main)
                x_gen = self.generator(batch_size=self.nb, device=self.device, **gkw)
                main.wait_stream(side)
            else:
                with torch.no_grad():
                    f_dat = self.discriminator(x_data, **self.model_opts)
                # the real-data features are final here: start their all-gather now, it overlaps the
                # generator forward and the second critic pass
                with self._timed("allgather_early"):
                    pending = (parallel.all_gather_rows_async(f_dat)
                               if self.collectives and (self.scope == "global" or self.world == 1) else None)
                x_gen = self.generator(batch_size=self.nb, device=self.device, **gkw)
            # only the generator's variables are differentiated in this step (train.py:112): run the critic with
            # its variables frozen, so that its layers skip their weight gradients (autograd's needs_input_grad
            # follows requires_grad, not the `inputs` list of autograd.grad) and only propagate d/dx
            with _frozen(self.disc_params):
                f_gen = self.discriminator(x_gen, **self.model_opts)
            with self._timed("match_total"):
                g_gen, _g_dat, dist, ent = self._match(f_gen.detach(), f_dat, pending, need_dat=False)
            if self.gen_buckets is not None:
                self.gen_buckets.arm()
            grads = self._backward(f_gen, self.gen_params, g_gen)                                 # train.py:112
            with self._timed("allreduce"):
                grads = (self.gen_buckets.finish() if self.gen_buckets is not None
                         else parallel.allreduce_sum_(list(grads)))
            if apply_updates:
                self._optimise(self.gen_optimizer, grads, a.learning_rate_gen, critic=False)      # train.py:142, 223
        return dist, ent, grads

    @torch.no_grad()
    def sample(self, n, ema=False):
        return self.generator(batch_size=n, ema=self.ema if ema else None, device=self.device, **self.model_opts)

    # ---------------------------------------------------------------- checkpoint (train.py:60,190-193,275-277)
    def state_dict(self, full=True):
        """Variables under the reference's names ('discriminator/conv2d_0/V', ...) + the step counter --
        what the reference's Saver(trainable_variables) keeps (train.py:60) -- and, with `full` (default;
        SURVEY 8f-3), what it omits: the optimisers' moments and step count and the EMA shadows, so
        that a resumed run continues bit for bit instead of restarting Adam at t = 1 with an
        untrained EMA generator.  Format: a torch-pickled dict of CPU tensors (not a TF checkpoint)."""
        sd = {"step_counter": self.step_counter}
        for t in (self.discriminator, self.generator):
            sd.update({k: v.detach().cpu() for k, v in t.named_variables().items()})
        if full:
            sd["__optim__"] = {"gen": self.gen_optimizer.state_dict(), "disc": self.disc_optimizer.state_dict()}
            names = list(self.generator.named_variables())
            sd["__ema__"] = {n: self.ema.average(p).detach().cpu() for n, p in zip(names, self.gen_params)}
        return sd

    def load_state_dict(self, sd):
        from . import ops
        if self.graphs is not None:
            self.graphs.drop("checkpoint loaded (the optimisers' moment buffers are new tensors)", recapture=True)
        with torch.no_grad():
            for t in (self.discriminator, self.generator):
                for k, v in t.named_variables().items():
                    v.copy_(sd[k].to(v.device))
            names = list(self.generator.named_variables())
            ema = sd.get("__ema__")
            for n, p in zip(names, self.gen_params):
                # a weights-only checkpoint (the reference's kind): the shadows restart AT the loaded
                # weights, as tf.train.ExponentialMovingAverage initialises them -- never at the random init
                src = ema[n].to(p.device) if ema is not None else p.detach()
                self.ema.average(p).copy_(src)
        opt = sd.get("__optim__")
        if opt is not None:
            self.gen_optimizer.load_state_dict(opt["gen"])
            self.disc_optimizer.load_state_dict(opt["disc"])
        ops.bump_weights_epoch()
        self.step_counter = int(sd.get("step_counter", 0))

    def check_finite(self):
        """Synchronising sanity check of the last step's scalars (the persistent Sinkhorn kernel
        reports a failed -- not co-resident -- launch by poisoning entropy and distance with NaN).
        The hot loop never calls this; train.py does once per logging interval."""
        for k in ("distance", "entropy"):
            v = self.last.get(k)
            if v is not None and not bool(torch.isfinite(v)):
                raise FloatingPointError(f"non-finite matching {k} in step {self.step_counter - 1}: the Sinkhorn "
                                         "kernel failed (device shared / partitioned?) or the model diverged")

    def close(self):
        """Uninstall the gradient-bucket hooks from the (module-level, shared) variables."""
        for b in (self.disc_buckets, self.gen_buckets):
            if b is not None:
                b.remove()
        self.disc_buckets = self.gen_buckets = None
        if self.graphs is not None:
            self.graphs.drop("trainer closed")
            self.graphs = None


class GraphedSteps:
    """Whole training steps captured as hipGraphs and replayed (the launch-bound inner loop of the reference's
    `sess.run`, train.py:207-226: ~130 launches per DCGAN step, ~640 per DenseNet step, each behind Python, ctypes and
    the autograd engine -- 17 - 23 % of a DenseNet step was idle time between kernels in round 4).

    One graph per step KIND, because the step's launch sequence depends on which cached operands are valid:
        "disc"  critic step (phase 0 of the period nr_gen_per_disc + 1);
        "gen1"  the generator step right after it: the critic's normalised weights and Winograd-domain filters are
                recomputed here (the critic just changed);
        "gen"   every later generator step of the period: reads them.
    The first period always runs eagerly (workspaces, function attributes, optimiser state, device tables are created
    there); from then on a kind is captured the first time it is due -- which is the order gen1, gen, disc -- and
    replayed afterwards.  What crosses graph boundaries: the critic's cached operands (written by "gen1", read by "gen"
    and "disc": captured after it, so they hold its addresses) and the parameter / moment / EMA buffers (static).  The
    generator's cached operands never cross: every kind recomputes them (it changed in the previous step; "gen1"
    invalidates them itself, the critic step before it does not touch the generator).

    What changes from step to step comes from device memory: the data batch (copied into a static buffer), the RNG
    (torch's graph-safe generator state) and Adam's bias corrections (`coef_dev`, written before each replay with the
    values the eager step would pass: otgan_adam_coefficients).  A replayed step is therefore the SAME arithmetic as
    the eager one -- tests/test_step_graph_gpu.py asserts bit-identical parameters after a period of each.  amax
    records used inside a capture come from a pool allocated (zeroed) inside it, so every replay starts from zeroed
    records like the eager step does.  Any failure to capture disables the graphs with a warning; the eager step is
    always available.

    DEFAULT FOR DENSENET, off for DCGAN (`--step_graph [0|1]`, OTGAN_STEP_GRAPH=0/1 override).  Measured in round 6 on one box
    (profiles/r06_step_graph_two_stream_ab.txt; eager / replay with one stream / replay with both streams' chains captured):
    DenseNet 27.6 - 29.6 / 27.2 / 25.5 ms per step -- the step is ~600 launches, each behind Python, ctypes and the autograd
    engine, and the eager figure moves by 2 ms from run to run with the host; DCGAN 8.52 - 8.56 / 8.90 / 8.60 ms -- ~130 launches,
    not launch-bound, and the runtime's graph launch orders every node behind its predecessors where stream launches let a
    kernel's first workgroups start under the previous kernel's tail.  (Round 5 measured replays 3 - 6 % SLOWER on both models:
    every replay then recomputed the critic's normalised weights and Winograd filters -- `_bump_written` -- and the capture
    held one stream.)"""

    def __init__(self, model):
        self.m = model
        self.graphs = {}          # kind -> (CUDAGraph, dist, ent)
        self.x = None             # static input batch
        self.stream = None
        self.warm = 0             # eager steps seen
        self.dead = None          # reason, once disabled

    def drop(self, why, recapture=False):
        self.graphs = {}
        self.m._d_weights_in_graph = False
        self.warm = 0
        if not recapture:
            self.dead = why

    def _kind(self, phase):
        return "disc" if phase == 0 else ("gen1" if phase == 1 else "gen")

    def run(self, x_data, phase):
        """Replay (capturing first if due) the step of `phase`; None = run it eagerly."""
        from . import ops
        m = self.m
        period = m.args.nr_gen_per_disc + 1
        if self.dead is not None or _lib_prof_on():
            return None                   # (profiled passes bracket library launches: eager)
        if self.warm < period:            # the first period: eager
            self.warm += 1
            return None
        kind = self._kind(phase)
        if kind not in self.graphs:
            # capture order: "gen1" first (right after an eager or replayed critic step), then the kinds that read its
            # operands
            if kind != "gen1" and period > 1 and not m._d_weights_in_graph:
                return None
            try:
                self._capture(x_data, kind)
            except Exception as e:          # never lose a run over the optimisation
                import warnings
                warnings.warn(f"otgan_amd: step graph capture failed ({type(e).__name__}: {e}); continuing with eager steps")
                self.drop(f"capture failed: {e}")
                return None
        elif kind != "gen1" and period > 1 and not m._d_weights_in_graph:
            return None                     # an eager step refreshed the critic elsewhere: wait for the next "gen1"
        g, dist, ent = self.graphs[kind]
        opt = m.disc_optimizer if kind == "disc" else m.gen_optimizer
        self.x.copy_(x_data)
        if getattr(opt, "coef_dev", None) is not None:
            opt.write_coefficients()
        g.replay()
        # the host-side bookkeeping of the step
        opt.t += 1.0
        m.step_counter += 1
        m._d_weights_in_graph = kind != "disc"
        self._bump_written(kind)
        return {"kind": "disc" if kind == "disc" else "gen", "distance": dist, "entropy": ent}

    def _bump_written(self, kind):
        """The host-side effect of the replayed optimiser step: invalidate the cached operands of the network it updated
        (and of the EMA shadows a generator step moves) -- what ops.adam_step_gather does in an eager step.  The OTHER
        network's cached operands stay valid: they are the tensors a graph of this period wrote ("gen1" for the critic), which
        is what lets "gen" and "disc" be captured reading them.  (Until round 6 this bumped the global epoch: every kind was
        then captured with the critic's cache invalid and recomputed the critic's normalised weights and Winograd filters
        in every replay -- more work than the eager step it was measured against)"""
        from . import ops
        m = self.m
        ts = list(m.disc_params if kind == "disc" else m.gen_params)
        if kind != "disc":
            ts += [m.ema.average(p) for p in m.gen_params]
        seen = set()
        for t in ts:
            key = t.untyped_storage().data_ptr()
            if key not in seen:
                seen.add(key)
                ops.bump_weights_epoch(t)

    def _capture(self, x_data, kind):
        from . import ops
        m = self.m
        if self.x is None:
            self.x = torch.empty_like(x_data)
            self.stream = torch.cuda.Stream(device=m.device)
        for opt in (m.gen_optimizer, m.disc_optimizer):
            if hasattr(opt, "ensure_state"):
                opt.ensure_state()
                if opt.coef_dev is None:
                    opt.coef_dev = torch.ones(2, dtype=torch.float32, device=m.device)
        self.x.copy_(x_data)
        if kind == "gen1":
            # the generator's cached operands must not cross graphs (see the class comment): recompute them here
            ops.bump_weights_epoch(m.gen_params[0])
        saved_t = (m.gen_optimizer.t, m.disc_optimizer.t)
        ops.reset_amax_pool()
        g = torch.cuda.CUDAGraph()
        try:
            m.gen_optimizer.capturing = m.disc_optimizer.capturing = True      # Adam reads its bias corrections from coef_dev
            # the second stream's chains are captured with the step: the side stream joins the capture where the step forks
            # (wait_stream on the capturing stream) and every fork is joined before the step ends, so the graph holds two
            # parallel branches wherever the eager step runs two streams
            ops.SIDE_STREAM = m._side_stream if m.fork_wgrad else None
            with torch.cuda.graph(g, stream=self.stream):
                dist, ent, _ = m._step_body(self.x, "disc" if kind == "disc" else "gen")
        finally:
            ops.SIDE_STREAM = None
            m.gen_optimizer.capturing = m.disc_optimizer.capturing = False
            m.gen_optimizer.t, m.disc_optimizer.t = saved_t      # (the capture only recorded the launches)
            ops.reset_amax_pool()                                 # eager code must not draw from the graph's pool
        self.graphs[kind] = (g, dist, ent)


def rank_log_kernel_slices(rank, world, f_gen, f_dat, fa, fb, lam):
    """The three [nb, N] log-kernel row slices rank `rank` of `world` computes, exactly the reference's
    row sharding of the cost GEMMs (matching.py:29-39): a rank of the first half owns rows of a1 / b1 and
    makes its rows of (a1,a2) (a1,b1) (a1,b2); a rank of the second half owns rows of a2 / b2 and makes
    its rows of (b2,b1) (a2,b1) (a2,b2).  f_gen / f_dat: the rank's own [nb, D] features; fa / fb: the
    gathered global features, flat [2N, D] (halves taken as views) or as shard lists.  Returns [3, nb, N]."""
    def halves(f):      # a flat [2N, D] tensor (views, no copy) or the reference's shard list
        if torch.is_tensor(f):
            return f[:f.shape[0] // 2], f[f.shape[0] // 2:]
        return torch.cat(f[:len(f) // 2], 0), torch.cat(f[len(f) // 2:], 0)
    a2 = halves(fa)[1]
    b1, b2 = halves(fb)
    # one launch for the rank's three slices; blocks named twice (the rank's own rows, b1) are staged once
    if rank < world // 2:   # my rows belong to a1 / b1:  p0 (a1,a2), p2 (a1,b1), p3 (a1,b2)
        return matching.cost_log_kernels([f_gen, f_gen, f_gen], [a2, b1, b2], lam)
    # my rows belong to a2 / b2:  p1 (b2,b1), p4 (a2,b1), p5 (a2,b2)
    return matching.cost_log_kernels([f_dat, f_gen, f_gen], [b1, b1, b2], lam)


def rank_stack_ok(nb, allg):
    """Does the one-split-per-step path (matching.FeatureStack) take a rank with `nb` rows of the gathered [2N, D] features?
    (The split-precision matching engine's shapes: N >= 256, D % 32 == 0, a rank's row slice at least one 256-row tile.)"""
    N, D = allg.shape[0] // 2, allg.shape[1]
    return nb >= 256 and nb % 32 == 0 and N % nb == 0 and matching.FeatureStack.supported(N, D)


def rank_matching_stack(rank, world, nb, allg, alld, lam, iters, need_dat, gather=None):
    """What rank `rank` of `world` runs per step in the global matching scope, with ONE split of the gathered features
    (reference utils/matching.py:29-39 row sharding of the cost GEMMs, :64-83 plan application): the feature stack filled
    where this rank reads it, its three [nb, N] cost row slices from the stack, the slices of all ranks gathered
    (`gather`: [3, nb, N] -> [world, 3, nb, N]; None = precomputed by the caller as `gather(mine)` is not needed for one
    rank's timing) and assembled, the six Sinkhorn problems, the plans applied to the rank's rows from the same stack.
    -> (grad_gen, grad_dat or None, entropy, distance)."""
    N = allg.shape[0] // 2
    ranges, own_gen, own_dat = matching.FeatureStack.rank_plan(rank * nb, nb, N, need_dat)
    st = matching.FeatureStack(allg, alld, ranges)
    if rank < world // 2:       # p0 (a1,a2), p2 (a1,b1), p3 (a1,b2)
        mine = st.cost_slices([own_gen] * 3, [3 * N, N, 2 * N], nb, lam)
    else:                       # p1 (b2,b1), p4 (a2,b1), p5 (a2,b2)
        mine = st.cost_slices([own_dat, own_gen, own_gen], [N, N, 2 * N], nb, lam)
    allk = gather(mine) if callable(gather) else gather
    K = assemble_log_kernels(allk, world)
    return st.rows_grad(lam, iters, (rank * nb, nb), K, need_b=need_dat)


def rank_single_log_kernel_slices(f_gen, f_dat, allg, alld, lam):
    """--single_batch: the three [nb, n] log-kernel row slices one rank computes -- its rows of a-a, b-b and a-b, exactly the
    reference's per-tower GEMMs (matching.py:99-104); the 999 on the a-a / b-b diagonals (:107-108) is added after the
    slices are assembled.  f_gen / f_dat: the rank's own [nb, D] features; allg / alld: the gathered [n, D] arrays.
    Returns [3, nb, n]."""
    return matching.cost_log_kernels([f_gen, f_dat, f_gen], [allg, alld, alld], lam)


def assemble_single_log_kernels(allk, lam):
    """[world, 3, nb, n] all-gathered slices -> [3, n, n] (a-a, b-b, a-b) with -lambda*999 on the a-a / b-b diagonals."""
    n = allk.shape[3]
    K = allk.permute(1, 0, 2, 3).reshape(3, n, n).contiguous()
    K[0].diagonal().add_(-float(lam) * 999.0)
    K[1].diagonal().add_(-float(lam) * 999.0)
    return K


_ASSEMBLE_IDX = {}


def assemble_log_kernels(allk, world):
    """[world, 3, nb, N] all-gathered slices -> the six [N, N] log-kernels in the reference's problem order
    a1a2, b2b1, a1b1, a1b2, a2b1, a2b2 (matching.py:41-43)."""
    W2 = world // 2
    N = allk.shape[3]
    # problem p = slice js[p] of the ranks of half hs[p]: ONE gather (three copies as permute / reshape / stack before)
    key = (allk.device, "assemble")
    idx = _ASSEMBLE_IDX.get(key)
    if idx is None:
        idx = (torch.tensor([0, 1, 0, 0, 1, 1], device=allk.device), torch.tensor([0, 0, 1, 2, 1, 2], device=allk.device))
        _ASSEMBLE_IDX[key] = idx
    a5 = allk.reshape(2, W2, 3, allk.shape[2], N)
    return a5[idx[0], :, idx[1]].reshape(6, N, N)      # problems 0, 2, 3 from the first half's ranks; 1, 4, 5 from the second's


def default_args(**over):
    """The reference's flag defaults (train.py:14-33) plus the added ones."""
    import argparse
    d = dict(seed=1, batch_size=625, learning_rate_disc=0.0003, learning_rate_gen=0.0003,
             data_dir='/home/tim/data', save_dir='/local_home/tim/med_gan', optimizer='adam',
             nonlinearity='crelu', nr_gpu=8, nr_gen_per_disc=5, sinkhorn_lambda=500.,
             nr_sinkhorn_iter=500, single_batch=False, train_disc_against_ema=False, model='dcgan',
             load_params=False, model_name='med_gan_params-2399', no_sinkhorn=False,
             image_size=32, matching_scope='global', synthetic=False, max_steps=0, save_every=200,
             synthetic_size=50000, data_dependent_init=False, eval_every=100, eval_samples=50000,
             inception_model='', ranks=0, step_graph=None)
    d.update(over)
    return argparse.Namespace(**d)


def smoke_step(device):
    """One tiny critic step + one generator step (used by __graft_entry__.smoke())."""
    args = default_args(batch_size=4, nr_gpu=2, nr_sinkhorn_iter=10, nr_gen_per_disc=1)
    m = OTGAN(args, device)
    x = torch.rand(m.nb, 32, 32, 3, device=device) * 2 - 1
    r0 = m.step(x)
    r1 = m.step(x)
    assert r0["kind"] == "disc" and r1["kind"] == "gen"
    for r in (r0, r1):
        assert torch.isfinite(r["distance"]).item() and torch.isfinite(r["entropy"]).item()
    print("smoke ok: train steps", float(r0["distance"]), float(r1["distance"]))
