"""Annealed reverse-SDE sampler driving the HIP kernels.

Mirror of ``WeightedSDEIntegrator`` (pita/src/models/components/sde_integration.py:48-470):
same constructor, ``integrate_sde(x1, energy_function, annealing_factor_schedule,
inverse_temperature, annealing_factor_score, resampling_interval)`` and the same 5-tuple result.

How the work is laid out on MI355X (differences from the reference are deliberate):
  * the per-step scalars (h, g^2, gamma, EDM coefficients, dt) are computed once on the HOST in the
    reference's fp32 op order and uploaded as a [N, 16] table;
  * with the HIP EGNN backbone the whole trajectory between two resampling events is ONE kernel
    launch (pita_egnn_sampler_run): walkers stay in registers, no per-step launch, no per-step
    all_gather, no per-step device->host copy of SDETerms (reference :248-258,:289);
  * walkers are sharded over ranks by contiguous slices exactly like :227-233, and gathered ONCE
    at the end (RCCL all_gather); a resampling event is global like the reference's, but only the
    log-weights are gathered -- the walkers move in one all_to_all_single that carries each rank the
    distinct parents it needs (_Comm.exchange_rows);
  * noise is Philox4x32-10 keyed by (seed, GLOBAL walker index, step): the result does not depend
    on the number of GPUs.  ``noise=`` injects recorded normals for parity tests.
``sde_terms_all`` has N entries like the reference's (:150,:212).  By default each field is a ``TermStats`` (sum,
sum of squares, count over the global batch, reduced on the device) that answers ``.mean()`` / ``.std()`` -- the only
things the reference's callers ever ask of these tensors (energytemp_module.py:938-945,1132-1143);
``record_terms=True`` materialises the full per-step tensors instead (per-step launch path).
"""
import math

import numpy as np
import torch

from . import _lib
from .data_utils import remove_mean
from .sdes import SDETerms, TermStats, VEReverseSDE  # noqa: F401
from .utils import gather_rows, sample_cat_sys


def _scalar(v):
    return float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v)


def _quantile_clamp(a, q):
    """clamp(a, max=quantile(a, q)) over the whole vector through the K11 kernel (sde_integration.py:179)."""
    a = _lib.dev_tensor(a, "a").clone()
    _lib.check(_lib.lib().pita_quantile_clamp(a.data_ptr(), a.shape[0], a.shape[0], float(q), _lib.stream_ptr(a.device)),
               "pita_quantile_clamp")
    return a


_STEP_TABLES = {}  # key -> table; the last few (schedule, grid) combinations of this process


def _schedule_key(obj):
    """Hashable identity of a schedule by VALUE (class + plain-number attributes); None when it carries anything else."""
    try:
        attrs = vars(obj)
    except TypeError:  # no __dict__ (slots, builtins): not cacheable by value
        return None
    items = []
    for k, v in sorted(attrs.items()):
        if isinstance(v, (bool, int, float, str, type(None))):
            items.append((k, v))
        else:
            return None
    return (type(obj).__module__, type(obj).__qualname__, tuple(items))


def build_step_table(noise_schedule, annealing_factor_schedule, times, dt, diffusion_scale, inverse_temperature):
    """[N, 16] float32 host table of per-step scalars in the reference's fp32 op order
    (score_net.py:26-29; sdes.py:119-122,140,250; sde_integration.py:347).

    The N rows are computed one 0-dim tensor at a time, as the reference computes them (a vectorised evaluation may
    differ in the last bit of a ``pow``): 65 ms for 1 000 steps on the GPU box's host -- 6 % of a whole 1 000-step LJ13
    trajectory and a third of a DW4 one.  The table only depends on the two schedules' parameters and the grid, so the
    last few are kept by value and a repeated ``integrate_sde`` with the same settings gets a copy."""
    ks, ka = _schedule_key(noise_schedule), _schedule_key(annealing_factor_schedule)
    key = None
    if ks is not None and ka is not None:
        tt = torch.as_tensor(times, dtype=torch.float32).detach().cpu().contiguous()
        key = (ks, ka, tt.numpy().tobytes(), float(dt), float(diffusion_scale), float(inverse_temperature))
        hit = _STEP_TABLES.get(key)
        if hit is not None:
            return hit.clone()
    tab = _build_step_table(noise_schedule, annealing_factor_schedule, times, dt, diffusion_scale, inverse_temperature)
    if key is not None:
        if len(_STEP_TABLES) >= 8:
            _STEP_TABLES.pop(next(iter(_STEP_TABLES)))
        _STEP_TABLES[key] = tab.clone()
    return tab


def _build_step_table(noise_schedule, annealing_factor_schedule, times, dt, diffusion_scale, inverse_temperature):
    N = len(times)
    tab = torch.zeros(N, _lib.STEP_STRIDE, dtype=torch.float32)
    sqrt_dt = np.sqrt(dt)
    for k in range(N):
        t = times[k]  # 0-dim fp32 CPU tensor
        ht = noise_schedule.h(t)
        g = noise_schedule.g(t)
        tab[k, _lib.ST_CS] = 1 / (1 + ht)
        c_in = 1 / (1 + ht) ** 0.5
        tab[k, _lib.ST_CIN] = c_in
        tab[k, _lib.ST_COUT] = ht**0.5 * c_in
        tab[k, _lib.ST_CNOISE] = (1 / 8) * torch.log(ht)
        tab[k, _lib.ST_H] = ht
        tab[k, _lib.ST_G2] = g.pow(2)
        tab[k, _lib.ST_GAMMA] = _scalar(annealing_factor_schedule.gamma(t))
        tab[k, _lib.ST_DT] = dt
        tab[k, _lib.ST_NOISE_SCALE] = diffusion_scale * g
        tab[k, _lib.ST_SQRT_DT] = sqrt_dt
        tab[k, _lib.ST_BETA] = float(inverse_temperature)
    return tab


class _Comm:
    """Rank / world / all_gather, from the Lightning module the reference passes
    (sde_integration.py:227-229,248) or from torch.distributed (RCCL) when that is initialised."""

    def __init__(self, lightning_module):
        self.lm = lightning_module
        if lightning_module is not None and getattr(lightning_module, "trainer", None) is not None:
            self.world = int(lightning_module.trainer.world_size)
            self.rank = int(lightning_module.trainer.global_rank)
        elif torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size()
            self.rank = torch.distributed.get_rank()
        else:
            self.world, self.rank = 1, 0

    def all_gather(self, x):
        if self.world == 1:
            return x
        if self.lm is not None and hasattr(self.lm, "all_gather") and not torch.distributed.is_initialized():
            return self.lm.all_gather(x).reshape(-1, *x.shape[1:])
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        torch.distributed.all_gather_into_tensor(out, x.contiguous())
        return out

    def all_reduce_sum(self, x):
        if self.world == 1:
            return x
        return self.all_gather(x[None]).reshape(self.world, *x.shape).sum(0)

    def shared_uniform(self, u):
        """The resampling uniform of this event, identical on every rank (the reference draws it from each rank's CPU
        generator, utils.py:111-120, and relies on identically seeded ranks; the exchange below needs agreement)."""
        if u is None:
            u = torch.rand(size=(1,), dtype=torch.float64)
        u = torch.as_tensor(u, dtype=torch.float64).reshape(-1)[:1].clone()
        if self.world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
            if torch.distributed.get_backend() == "nccl":
                ud = u.cuda()
                torch.distributed.broadcast(ud, 0)
                u = ud.cpu()
            else:
                torch.distributed.broadcast(u, 0)
        return u

    def exchange_rows(self, x, ids, Bl):
        """Rows ``ids[rank*Bl:(rank+1)*Bl]`` of the batch whose shards are the ranks' ``x`` [Bl, D]; ``ids`` [world*Bl]
        are the parents of a global systematic resampling: identical on every rank, non-decreasing up to the cyclic
        rotation by the event's uniform (utils.py:111-120: ``(u + i / bs) % 1``).  Instead of
        all-gathering every walker to every rank (the reference, sde_integration.py:248-258), each rank sends a
        destination only the DISTINCT parents it holds that the destination needs (one RCCL all_to_all_single with
        uneven splits); repeated parents are expanded locally."""
        W, r = self.world, self.rank
        if W == 1:
            return gather_rows(x, ids)
        if Bl == 0:  # fewer walkers than ranks: every shard is empty (sde_integration.py:227)
            return x
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return gather_rows(self.all_gather(x), ids)[r * Bl:(r + 1) * Bl].clone()  # Lightning's all_gather only
        n = W * Bl
        first = torch.ones(n, dtype=torch.bool, device=ids.device)
        first[1:] = ids[1:] != ids[:-1]
        first[::Bl] = True  # a destination's first walker always needs its parent
        src = torch.div(ids, Bl, rounding_mode="floor")
        dest = torch.div(torch.arange(n, device=ids.device), Bl, rounding_mode="floor")
        counts = torch.bincount((dest * W + src)[first], minlength=W * W).reshape(W, W).tolist()  # [dest][src]
        send = x[ids[first & (src == r)] - r * Bl].contiguous()  # grouped by destination, ascending
        recv = torch.empty((sum(counts[r]), x.shape[1]), device=x.device, dtype=x.dtype)
        in_split, out_split = [counts[d][r] for d in range(W)], counts[r]
        if torch.distributed.get_backend() != "nccl" and x.is_cuda:  # gloo rehearsal: through the host
            rc = torch.empty(recv.shape, dtype=x.dtype)
            torch.distributed.all_to_all_single(rc, send.cpu(), output_split_sizes=out_split, input_split_sizes=in_split)
            recv = rc.to(x.device)
        else:
            torch.distributed.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split)
        self.rows_received = getattr(self, "rows_received", 0) + sum(counts[r]) - counts[r][r]
        # what this rank put on the wire (rows to other ranks; the split it kept for itself never leaves the device)
        self.rows_sent = getattr(self, "rows_sent", 0) + sum(in_split) - in_split[r]
        self.last_splits = {"sent_to": in_split, "received_from": out_split}
        # the received rows are ordered by (source rank, position in my slice); my slice's runs are ordered by position
        fl = first[r * Bl:(r + 1) * Bl]
        jl = torch.nonzero(fl).reshape(-1)
        order = torch.argsort(src[r * Bl:(r + 1) * Bl][jl] * Bl + jl)
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.numel(), device=order.device)
        return recv[pos[torch.cumsum(fl.to(torch.int64), 0) - 1]]


def _count_distinct(ids):
    """Number of distinct parents of a systematic resampling, as a 0-dim DEVICE tensor (no host synchronisation).  The
    ids are non-decreasing up to the cyclic rotation by the event's uniform (utils.py:111-120), so every distinct parent
    is one cyclic run: distinct = max(1, number of positions whose cyclic predecessor differs).  The reference takes
    ``len(np.unique(choice))`` on the host every step (sde_integration.py:295)."""
    if ids.is_cuda:  # one launch (pita_count_runs) instead of roll + != + cast + sum + clamp
        ids = ids.contiguous()
        out = torch.empty((), device=ids.device, dtype=torch.int64)
        _lib.check(_lib.lib().pita_count_runs(ids.data_ptr(), ids.numel(), out.data_ptr(), _lib.stream_ptr(ids.device)),
                   "pita_count_runs")
        return out
    return (ids != torch.roll(ids, 1)).sum().clamp_min(1)  # host tensors: the gloo / dry-run paths


def _host_counts(counts):
    """The list ``num_unique_idxs`` of the reference (python ints): one transfer for all the events of a run."""
    dev = [(k, c) for k, c in enumerate(counts) if isinstance(c, torch.Tensor)]
    if dev:
        vals = torch.stack([c for _, c in dev]).tolist()
        for (k, _), v in zip(dev, vals):
            counts[k] = int(v)
    return counts


def _terms_from_stats(st4, st8, n_elem, n_walk, computed, debiased):
    """N light SDETerms from the per-step moment buffers (host copies happen once, here).  st4: [N, 4] drift_X /
    diffusion moments; st8: [N, 8] drift_A, divergence_score, cross_term, dUt_dt moments (debiased regime only)."""
    h4 = st4.cpu().tolist()
    h8 = st8.cpu().tolist() if st8 is not None else None
    out = []
    for k, r in enumerate(h4):
        ne, nw = (n_elem, n_walk) if computed[k] else (0, 0)
        t = SDETerms(drift_X=TermStats(r[0], r[1], ne), drift_A=TermStats(0.0, 0.0, nw),
                     diffusion=TermStats(r[2], r[3], ne))
        if debiased:
            q = h8[k]
            t.drift_A = TermStats(q[0], q[1], nw)
            t.divergence_score, t.cross_term, t.dUt_dt = (TermStats(q[2], q[3], nw), TermStats(q[4], q[5], nw),
                                                          TermStats(q[6], q[7], nw))
        out.append(t)
    return out


class WeightedSDEIntegrator:
    def __init__(self, sde, num_integration_steps, start_resampling_step, end_resampling_step, lightning_module=None,
                 partial_annealing_factor_schedule=None, reverse_time=True, diffusion_scale=1.0, time_range=1.0,
                 resampling_interval=-1, num_negative_time_steps=100, post_mcmc_steps=100, adaptive_mcmc=False,
                 batch_size=None, no_grad=True, resample_at_end=False, dt_negative_time=1e-4, do_langevin=False,
                 should_mean_free=True, seed=0, record_terms=False, verbose=False):
        self.sde = sde
        self.num_integration_steps = num_integration_steps
        self.start_resampling_step, self.end_resampling_step = start_resampling_step, end_resampling_step
        self.reverse_time = reverse_time
        self.diffusion_scale = diffusion_scale
        self.resampling_interval = resampling_interval
        self.time_range = time_range
        self.num_negative_time_steps = num_negative_time_steps
        self.post_mcmc_steps = post_mcmc_steps
        self.adaptive_mcmc = adaptive_mcmc
        self.dt_negative_time = dt_negative_time
        self.batch_size = batch_size
        self.no_grad = no_grad
        self.resample_at_end = resample_at_end
        self.do_langevin = do_langevin
        self.lightning_module = lightning_module
        self.should_mean_free = should_mean_free
        self.start_time = time_range if reverse_time else 0.0
        self.end_time = time_range - self.start_time
        self.seed = seed
        self.record_terms = record_terms
        self.verbose = verbose
        self._runs = 0

    # ------------------------------------------------------------------ helpers
    def maybe_remove_mean(self, x, energy_function):
        if self.should_mean_free:
            return remove_mean(x, energy_function.n_particles, energy_function.n_spatial_dim)
        return x

    def _geometry(self, x, energy_function):
        n = getattr(energy_function, "n_particles", None)
        d = getattr(energy_function, "n_spatial_dim", None)
        if n is None or d is None:  # non-particle target (GMM): one "particle" of dimension D
            n, d = 1, x.shape[1]
        return int(n), int(d)

    def _backbone(self):
        sn = self.sde.score_net
        model = getattr(sn, "model", None)
        fused = model is not None and hasattr(model, "sampler_run") and not getattr(sn, "precondition_beta", False)
        return model if fused else None

    def _key(self, stream_id):
        return (int(self.seed) * 0x9E3779B97F4A7C15 + (self._runs << 8) + stream_id) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ A1 integrate_sde
    @torch.no_grad()
    def integrate_sde(self, x1, energy_function, annealing_factor_schedule, inverse_temperature=1.0,
                      annealing_factor_score=1.0, resampling_interval=None, noise=None, resample_u=None,
                      mala_noise=None, mala_uniforms=None):
        """``noise``: optional [N, B, D] device normals (global batch); ``resample_u``: optional
        iterable of float64 uniforms, one per resampling event; ``mala_noise`` [steps, B, D] / ``mala_uniforms``
        [steps, B]: the proposal normals and accept uniforms of the post-processing MALA chain (parity hooks;
        single rank)."""
        if resampling_interval is None:
            resampling_interval = self.resampling_interval
        N = self.num_integration_steps
        x1 = _lib.dev_tensor(x1, "x1")
        dev = x1.device
        comm = _Comm(self.lightning_module)
        mala_draws = (mala_noise, mala_uniforms)
        if comm.world > 1 and (mala_noise is not None or mala_uniforms is not None):
            # the hooks are shaped for ONE chain over the whole batch; refuse before the trajectory is integrated
            raise ValueError("mala_noise / mala_uniforms are single-rank parity hooks (world size is "
                             f"{comm.world}): every rank draws its own Philox stream instead")
        Bg = x1.shape[0]
        Bl = Bg // comm.world  # sde_integration.py:227
        off = comm.rank * Bl
        x = x1[off:off + Bl].clone()
        n, d = self._geometry(x, energy_function)
        mean_free = bool(self.should_mean_free)
        if self.batch_size is None:
            self.batch_size = Bg

        times = torch.linspace(self.start_time, self.end_time, N + 1)[:-1]
        dt = self.time_range / N
        tab_h = build_step_table(self.sde.noise_schedule, annealing_factor_schedule, times, dt, self.diffusion_scale,
                                 inverse_temperature)
        tab = tab_h.to(dev)
        if noise is not None:
            if tuple(noise.shape) != (N, Bg, x1.shape[1]):
                raise ValueError(f"noise has shape {tuple(noise.shape)}, expected {(N, Bg, x1.shape[1])}")
            noise = _lib.dev_tensor(noise, "noise")[:, off:off + Bl].contiguous()
        key = self._key(0)
        self._runs += 1
        u_iter = iter(resample_u) if resample_u is not None else None

        start = max(0, min(self.start_resampling_step, N))  # before `start` walkers do not move (:278-280)
        if start > 0 and mean_free:
            x = remove_mean(x, n, d)  # :148 is applied every step; idempotent
        did_resampling = resampling_interval != -1 and resampling_interval < N
        events = [] if resampling_interval == -1 else [
            s for s in range(start, min(self.end_resampling_step, N)) if (s + 1) % resampling_interval == 0]
        num_unique_idxs = [Bg] * N
        sde_terms_all = []
        st4 = None if self.record_terms else torch.zeros(N, 4, dtype=torch.float64, device=dev)
        model = self._backbone()
        if getattr(self.sde, "debias_inference", False):
            return self._integrate_debiased(x, comm, tab_h, times, noise, key, off, Bl, Bg, n, d, mean_free,
                                            energy_function, annealing_factor_schedule, inverse_temperature,
                                            resampling_interval, u_iter, mala_draws)

        s = start
        bounds = events + [N - 1]
        for stop in bounds:  # run steps s..stop inclusive, then (maybe) resample
            if stop < s:
                continue
            self._run_steps(model, x, tab, tab_h, s, stop + 1, noise, key, off, n, d, mean_free, inverse_temperature,
                            sde_terms_all, st4)
            s = stop + 1
            if stop in events:
                a = torch.zeros(comm.world * Bl, device=dev)  # drift_A = 0 in the not-debiased regime
                u = comm.shared_uniform(next(u_iter) if u_iter is not None else None)
                ids, _ = sample_cat_sys(a.shape[0], a, u)
                num_unique_idxs[stop] = _count_distinct(ids)
                x = comm.exchange_rows(x, ids, Bl)
                if mean_free:
                    x = remove_mean(x, n, d)
        # log-weights are identically zero here; a stride-0 view avoids N*B*4 bytes (reference stacks N copies)
        logweights = torch.zeros(1, Bg, device=dev).expand(N, Bg)
        if st4 is not None:
            # moments come from the world * Bl walkers that are integrated (Bg % world walkers are dropped, :227)
            sde_terms_all = _terms_from_stats(comm.all_reduce_sum(st4), None, comm.world * Bl * x1.shape[1],
                                              comm.world * Bl, [k >= start for k in range(N)], False)

        if self.resample_at_end and did_resampling:
            x, a_next, n_unique = self._resample_at_end(x, None, comm, times, energy_function, annealing_factor_schedule,
                                                        inverse_temperature, u_iter, off, Bl)
            logweights = torch.cat([logweights, a_next[None]])
            num_unique_idxs.append(n_unique)

        if self.num_negative_time_steps > 0:
            x = self.negative_time_descent(x, energy_function, walker_offset=off)
        acceptance_rate_list = []
        if self.post_mcmc_steps > 0:
            fn = self.metropolis_hastings_mala_adaptive if self.adaptive_mcmc else self.metropolis_hastings_mala
            kw = dict(dt_init=self.dt_negative_time) if self.adaptive_mcmc else {}
            x, acceptance_rate_list = fn(x, energy_function, return_acceptance_rate=True, walker_offset=off, comm=comm,
                                         noise=mala_draws[0], uniforms=mala_draws[1], **kw)
        x = self._gather_final(x, comm, Bl)  # X1: the only collective on the resampling-free path
        return x, logweights, _host_counts(num_unique_idxs), sde_terms_all, acceptance_rate_list

    # ------------------------------------------------------------------ debiased regime (per step; section 8(f) N1)
    def _integrate_debiased(self, x, comm, tab_h, times, noise, key, off, Bl, Bg, n, d, mean_free, energy_function,
                            gamma_schedule, beta, resampling_interval, u_iter, mala_draws=(None, None)):
        """Feynman-Kac weighted integration (sde_integration.py:131-152,214-297 with sdes.py:151-239): per step the
        drift of x and of the log-weights a per inference chunk, Euler-Maruyama update, window gates, global
        systematic resampling when due.  Resampling is global like the reference's: weights and walkers are
        all-gathered at every resampling event."""
        N = self.num_integration_steps
        dev = x.device
        L = _lib.lib()
        zero_a = torch.zeros(Bl, device=dev)  # never written to: every update of `a` makes a new tensor
        a = zero_a
        logweights, num_unique_idxs, sde_terms_all = [], [], []
        bs = self.batch_size or Bl
        st4 = st8 = None
        if not self.record_terms:
            st4 = torch.zeros(N, 4, dtype=torch.float64, device=dev)
            st8 = torch.zeros(N, 8, dtype=torch.float64, device=dev)
        st = _lib.stream_ptr(dev)
        for step in range(N):
            t = times[step]  # host scalar: the schedules' scalars (gamma, dgamma/dt) are then read without a device sync
            row = tab_h[step]
            if step < self.start_resampling_step:  # walkers frozen, weights zero (:278-280)
                a = zero_a
                logweights.append(comm.all_gather(a))
                num_unique_idxs.append(Bg)
                continue
            # one set of launches for the rank's whole shard; the quantile clamp keeps its per-chunk meaning
            terms = self.sde.f(t, x, beta, gamma_schedule, None, energy_function, resampling_interval, clamp_chunk=bs)
            drift = terms.drift_X.contiguous()
            nz = noise[step].contiguous() if noise is not None else None
            _lib.check(L.pita_em_step(x.data_ptr(), drift.data_ptr(), _lib.ptr(nz), Bl, n, d, float(row[_lib.ST_DT]),
                                      float(row[_lib.ST_NOISE_SCALE]), float(row[_lib.ST_SQRT_DT]), key, off, step, 0,
                                      _lib.ptr(st4[step]) if st4 is not None else 0, st), "pita_em_step")
            if st8 is not None:
                vs = [None if v is None else _lib.dev_tensor(v, "SDETerms field").contiguous()
                      for v in (terms.drift_A, terms.divergence_score, terms.cross_term, terms.dUt_dt)]
                for v in vs:  # the kernel reads Bl entries of each field
                    if v is not None and v.numel() != Bl:
                        raise ValueError(f"SDETerms field has {v.numel()} entries, expected one per walker ({Bl})")
                _lib.check(L.pita_moments4(*(_lib.ptr(v) for v in vs), Bl, st8[step].data_ptr(), st), "pita_moments4")
            a = torch.add(a, terms.drift_A, alpha=float(row[_lib.ST_DT]))  # a + drift_A dt in one launch
            if step >= self.end_resampling_step:
                a = zero_a
            n_unique = Bg
            due = not (resampling_interval == -1 or (step + 1) % resampling_interval != 0
                       or step >= self.end_resampling_step)
            if due:
                ag = comm.all_gather(a)
                u = comm.shared_uniform(next(u_iter) if u_iter is not None else None)
                ids, _ = sample_cat_sys(ag.shape[0], ag, u)
                n_unique = _count_distinct(ids)
                x = comm.exchange_rows(x, ids, Bl)
                a = zero_a
            if mean_free:
                x = remove_mean(x, n, d)
            logweights.append(comm.all_gather(a))
            num_unique_idxs.append(n_unique)
            if self.record_terms:
                sde_terms_all.append(terms)
        logweights = torch.stack(logweights)
        if st4 is not None:
            sde_terms_all = _terms_from_stats(comm.all_reduce_sum(st4), comm.all_reduce_sum(st8),
                                              comm.world * Bl * x.shape[1], comm.world * Bl,
                                              [k >= self.start_resampling_step for k in range(N)], True)
        did_resampling = resampling_interval != -1 and resampling_interval < N
        if self.resample_at_end and did_resampling:
            x, a_next, n_unique = self._resample_at_end(x, a, comm, times, energy_function, gamma_schedule, beta, u_iter,
                                                        off, Bl)
            logweights = torch.cat([logweights, a_next[None]])
            num_unique_idxs.append(n_unique)
        if self.num_negative_time_steps > 0:
            x = self.negative_time_descent(x, energy_function, walker_offset=off)
        acceptance_rate_list = []
        if self.post_mcmc_steps > 0:
            fn = self.metropolis_hastings_mala_adaptive if self.adaptive_mcmc else self.metropolis_hastings_mala
            kw = dict(dt_init=self.dt_negative_time) if self.adaptive_mcmc else {}
            x, acceptance_rate_list = fn(x, energy_function, return_acceptance_rate=True, walker_offset=off, comm=comm,
                                         noise=mala_draws[0], uniforms=mala_draws[1], **kw)
        return self._gather_final(x, comm, Bl), logweights, _host_counts(num_unique_idxs), sde_terms_all, acceptance_rate_list

    def _gather_final(self, x, comm, Bl):
        """All-gather of the final shards.  After MALA every shard is [its valid walkers, its set-aside walkers]; the
        reference runs the chain on the gathered batch and therefore returns [all valid, all set-aside] (quirk Q7):
        reorder to that when any rank set walkers aside."""
        xg = comm.all_gather(x)
        if comm.world == 1 or self.post_mcmc_steps <= 0:
            return xg
        nv = comm.all_gather(torch.tensor([getattr(self, "_last_mala_valid", Bl)], device=x.device, dtype=torch.int64)).tolist()
        if min(nv) == Bl:
            return xg
        head = [torch.arange(r * Bl, r * Bl + nv[r]) for r in range(comm.world)]
        tail = [torch.arange(r * Bl + nv[r], (r + 1) * Bl) for r in range(comm.world)]
        return xg[torch.cat(head + tail).to(xg.device)]

    def _resample_at_end(self, x, a, comm, times, energy_function, gamma_schedule, beta, u_iter, off, Bl):
        """End-of-trajectory reweighting (sde_integration.py:158-183): a_next = log p_target(x) + gamma E_theta(h(t_end), x)
        (+ the running log-weights a), clamped at its 0.9 quantile, then global systematic resampling.
        Returns (local slice of the resampled walkers, a_next over the global batch, number of distinct parents)."""
        t_end = times[min(self.end_resampling_step, self.num_integration_steps - 1)]
        # every rank evaluates its own shard (the reference evaluates the gathered batch on every rank); only the
        # log-weights are gathered, the walkers move in the exchange
        tb = torch.full((x.shape[0],), float(t_end), device=x.device)
        model_energy = self.sde.energy_net.forward_energy(self.sde.noise_schedule.h(tb), x, beta,
                                                          pin=bool(getattr(self.sde, "pin_energy", False)),
                                                          energy_function=energy_function, t=tb)  # :166-173
        a_next = energy_function(x) + model_energy * _scalar(gamma_schedule.gamma(t_end))
        if a is not None:
            a_next = a_next + a
        a_next = _quantile_clamp(comm.all_gather(a_next.contiguous()), 0.9)
        u = comm.shared_uniform(next(u_iter) if u_iter is not None else None)
        ids, _ = sample_cat_sys(a_next.shape[0], a_next, u)
        x = comm.exchange_rows(x, ids, Bl)
        return x, a_next, _count_distinct(ids)

    # ------------------------------------------------------------------ A2-A4 steps [s0, s1)
    def _run_steps(self, model, x, tab, tab_h, s0, s1, noise, key, off, n, d, mean_free, beta, sde_terms_all, st4=None):
        if s1 <= s0:
            return
        if model is not None and not self.record_terms and (not hasattr(model, "can_fuse") or model.can_fuse(n, d)):
            nz = noise[s0:s1].contiguous() if noise is not None else None
            model.sampler_run(x, tab[s0:s1].contiguous(), s1 - s0, noise=nz, seed=key, walker_offset=off, step0=s0,
                              remove_mean=mean_free, n_particles=n, n_dim=d,
                              stats_out=st4[s0:s1] if st4 is not None else None)
            return
        # per-step path: any backbone with forward(t, x, beta); drift through ScoreNet, update by pita_em_step
        L = _lib.lib()
        for k in range(s0, s1):
            row = tab_h[k]
            ht = torch.full((x.shape[0],), float(row[_lib.ST_H]), device=x.device)
            score = self.sde.score_net(ht, x, beta)
            drift = float(row[_lib.ST_GAMMA]) * (score * float(row[_lib.ST_G2]))
            drift = drift.contiguous()
            nz = noise[k].contiguous() if noise is not None else None
            _lib.check(L.pita_em_step(x.data_ptr(), drift.data_ptr(), _lib.ptr(nz), x.shape[0], n, d,
                                      float(row[_lib.ST_DT]), float(row[_lib.ST_NOISE_SCALE]),
                                      float(row[_lib.ST_SQRT_DT]), key, off, k, int(mean_free),
                                      _lib.ptr(st4[k]) if st4 is not None else 0, _lib.stream_ptr(x.device)),
                       "pita_em_step")
            if self.record_terms:
                sde_terms_all.append(SDETerms(drift_X=drift, drift_A=torch.zeros(x.shape[0], device=x.device)))

    # ------------------------------------------------------------------ A16 post-processing
    def negative_time_descent(self, x, energy_function, noise=None, walker_offset=0, fused=True):
        """x += F*dt (+ sqrt(2 dt) xi), remove_mean, repeated (sde_integration.py:353-360).  Pair targets run all
        steps in one launch (pita_lj_descent / pita_dw_descent, bit-identical to the per-step path below)."""
        n, d = self._geometry(x, energy_function)
        dt = float(self.dt_negative_time)
        key = self._key(1)
        L = _lib.lib()
        x = _lib.dev_tensor(x, "x").clone()
        ns, sq = (1.0 if self.do_langevin else 0.0), math.sqrt(2 * dt)
        nsteps = int(self.num_negative_time_steps)
        if noise is not None:
            noise = _lib.dev_tensor(noise, "noise")
            if tuple(noise.shape) != (nsteps,) + tuple(x.shape):
                raise ValueError(f"descent noise has shape {tuple(noise.shape)}, expected {(nsteps,) + tuple(x.shape)}")
        if fused and hasattr(energy_function, "fused_descent") and energy_function.fused_descent(
                x, nsteps, dt, ns, sq, seed=key, walker_offset=walker_offset, step0=0,
                remove_mean=self.should_mean_free, noise=noise) is not None:
            return x
        for k in range(nsteps):
            _, F = energy_function(x, return_force=True)
            nz = noise[k].contiguous() if noise is not None else None
            _lib.check(L.pita_em_step(x.data_ptr(), F.data_ptr(), _lib.ptr(nz), x.shape[0], n, d, dt, ns, sq, key,
                                      walker_offset, k, int(self.should_mean_free), 0, _lib.stream_ptr(x.device)),
                       "pita_em_step")
        return x

    def _mala(self, x, energy_function, adaptive, dt, noise=None, uniforms=None, return_acceptance_rate=False,
              walker_offset=0, comm=None, fused=True):
        """MALA with the reference's finite-mask semantics (:362-470): non-finite-logp walkers are set aside and
        re-appended AFTER the valid ones (order not preserved, quirk Q7).  Proposal, accept/reject and the step-size
        adaptation run as HIP kernels with the step size on the device: no host synchronisation inside the chain (the
        fused chains synchronise once, after the launch, when the acceptance rates -- which also say whether an adaptive
        chain's grid barrier held -- are copied to the host).
        With several ranks the acceptance count is all-reduced so the adaptation sees the global rate, as the
        reference (which runs the chain on the gathered batch on every rank) does."""
        n, d = self._geometry(x, energy_function)
        L = _lib.lib()
        x = _lib.dev_tensor(x, "x")
        dev = x.device
        logp_all = energy_function(x)
        valid = torch.isfinite(logp_all)
        x_valid, x_invalid = x[valid].contiguous(), x[~valid]
        logp = logp_all[valid].contiguous()
        Bv = x_valid.shape[0]
        # Philox keys follow the walkers' ORIGINAL global indices, not their position in the compacted batch
        ids = (torch.nonzero(valid).reshape(-1) + int(walker_offset)).contiguous() if Bv < x.shape[0] else None
        self._last_mala_valid = Bv
        total = Bv
        world = comm.world if comm is not None and torch.distributed.is_initialized() else 1
        if world > 1:
            tot = torch.tensor([Bv], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(tot)
            total = int(tot.item())
        steps = int(self.post_mcmc_steps)
        rates = torch.zeros(max(steps, 1), device=dev)
        done = 0
        if total > 0 and steps > 0:
            dt_dev = torch.tensor([dt], device=dev, dtype=torch.float64)
            count = torch.zeros(1, device=dev, dtype=torch.int32)
            x_prop = torch.empty_like(x_valid)
            key = self._key(2)
            # :397-398 centres through maybe_remove_mean, the adaptive variant (:458-461) unconditionally
            rm = int(bool(getattr(energy_function, "is_molecule", False)) and (adaptive or bool(self.should_mean_free)))
            st = _lib.stream_ptr(dev)
            # pair targets with a fused chain kernel: every step in ONE launch, walkers LDS-resident (bit-identical
            # to the loop below); the global acceptance rate of several ranks needs the per-step all-reduce below
            if fused and world == 1 and Bv > 0 and hasattr(energy_function, "fused_mala"):
                nz = uu = None
                if noise is not None:
                    nz = torch.stack([_lib.dev_tensor(noise[i], "noise") for i in range(steps)]).contiguous()
                    if tuple(nz.shape) != (steps,) + tuple(x_valid.shape):
                        raise ValueError(f"MALA noise has shape {tuple(nz.shape)}, expected {(steps,) + tuple(x_valid.shape)}")
                if uniforms is not None:
                    uu = torch.stack([_lib.dev_tensor(uniforms[i], "uniforms").reshape(-1) for i in range(steps)]).contiguous()
                    if tuple(uu.shape) != (steps, Bv):
                        raise ValueError(f"MALA uniforms have shape {tuple(uu.shape)}, expected {(steps, Bv)}")
                # an adaptive chain synchronises the grid every step and assumes an idle device; if a block's bounded wait
                # runs out (another stream or process held compute units) the launch marks the chain invalid with NaN
                # rates: restore the walkers and run the launch-per-kernel chain, which cannot fail this way
                backup = (x_valid.clone(), logp.clone(), dt_dev.clone()) if adaptive else None
                if energy_function.fused_mala(x_valid, logp, steps, dt_dev, adaptive, total, noise=nz, uniforms=uu, seed=key,
                                              walker_offset=walker_offset, walker_ids=ids, step0=0, remove_mean=rm,
                                              rates_out=rates) is not None:
                    # the validity check is the chain's ONE host synchronisation: the transfer of the rates the caller
                    # asked for (or of the final step size when it did not)
                    host_rates = rates[:steps].tolist() if return_acceptance_rate else (dt_dev.tolist() if adaptive else [])
                    if adaptive and any(math.isnan(r) for r in host_rates):
                        x_valid.copy_(backup[0])
                        logp.copy_(backup[1])
                        dt_dev.copy_(backup[2])
                        rates.zero_()
                        self._fused_mala_fallbacks = getattr(self, "_fused_mala_fallbacks", 0) + 1
                    else:
                        out = torch.cat([x_valid, x_invalid], dim=0)
                        return (out, host_rates) if return_acceptance_rate else (out, None)
            for i in range(steps):
                if Bv > 0:
                    _, grad = energy_function(x_valid, return_force=True)
                    nz = _lib.dev_tensor(noise[i], "noise") if noise is not None else None
                    if nz is not None and tuple(nz.shape) != tuple(x_valid.shape):
                        raise ValueError(f"MALA noise[{i}] has shape {tuple(nz.shape)}, expected {tuple(x_valid.shape)} "
                                         "(one row per walker with a finite target log-density)")
                    _lib.check(L.pita_mala_propose(x_valid.data_ptr(), grad.data_ptr(), x_prop.data_ptr(), _lib.ptr(nz),
                                                   Bv, n, d, dt_dev.data_ptr(), key, walker_offset, _lib.ptr(ids), i, st),
                               "pita_mala_propose")
                    logp_prop, grad_prop = energy_function(x_prop, return_force=True)
                    uu = _lib.dev_tensor(uniforms[i], "uniforms") if uniforms is not None else None
                    if uu is not None and uu.numel() != Bv:
                        raise ValueError(f"MALA uniforms[{i}] has {uu.numel()} entries, expected {Bv}")
                    _lib.check(L.pita_mala_accept(x_valid.data_ptr(), logp.data_ptr(), grad.data_ptr(), x_prop.data_ptr(),
                                                  logp_prop.data_ptr(), grad_prop.data_ptr(), _lib.ptr(uu), Bv, n, d,
                                                  dt_dev.data_ptr(), key, walker_offset, _lib.ptr(ids), i, rm,
                                                  count.data_ptr(), st), "pita_mala_accept")
                if world > 1:
                    torch.distributed.all_reduce(count)
                _lib.check(L.pita_mala_adapt(dt_dev.data_ptr(), count.data_ptr(), total, int(adaptive),
                                             rates[i:].data_ptr(), st), "pita_mala_adapt")
                done += 1
        out = torch.cat([x_valid, x_invalid], dim=0)
        return (out, rates[:done].tolist()) if return_acceptance_rate else (out, None)

    def metropolis_hastings_mala(self, x, energy_function, return_acceptance_rate=False, noise=None, uniforms=None,
                                 walker_offset=0, comm=None, fused=True):
        return self._mala(x, energy_function, False, float(self.dt_negative_time), noise, uniforms,
                          return_acceptance_rate, walker_offset, comm, fused)

    def metropolis_hastings_mala_adaptive(self, x, energy_function, dt_init, return_acceptance_rate=False, noise=None,
                                          uniforms=None, walker_offset=0, comm=None, fused=True):
        return self._mala(x, energy_function, True, float(dt_init), noise, uniforms, return_acceptance_rate,
                          walker_offset, comm, fused)
