"""Fused on-GPU relative-pose pipeline for batches of image pairs:

    SuperPoint x2 -> SuperGlue -> depth lift -> PnP-RANSAC (+ refinement)      [config sg_pnp_*]

Replaces, for config/matching/mapfree/sg_pnp_dptkitti.yaml, the reference's two-stage flow
(offline etc/feature_matching_baselines/compute.py:72-86 writing correspondences_SG.npz, then
submission.py:33-58 -> FeatureMatchingModel.forward (lib/models/matching/model.py:29-40) ->
PnPSolver.estimate_pose (pose_solver.py:184-235)) with one device-resident pass: matches never
leave HBM, poses come back as one small tensor per batch.  The npz wire format and the per-pair
plugin API stay available (matching/, wire.py) for drop-in use.
"""
import torch

from . import _lib, options
from .nets.superpoint import SuperPointHIP
from .nets.superglue import SuperGlueHIP
from .nets import weights as WT
from .solver_ops import PnPBatchSolver


ST_RANGE = 7            # include/mfr_hip.h MFR_ST_RANGE


class RangeGuard:
    """The f16x2 range guard of ONE pipeline object (include/mfr_hip.h mfr_f16x2_guard_bind, csrc/guard.h).

    The reference's networks are plain fp32 modules (matchers.py:50,105): they have no |x| <= 65504 precondition, the f16x2 kernels do.  Used as a
    context manager around the matcher stage: the device flag is cleared and bound, every f16x2 launch inside ORs 1 into it when one of its
    accumulators is non-finite (= an operand outside the range, or a non-finite operand), and `fold` turns a set flag into status ST_RANGE for the
    batch's pairs -- all on the device, no host synchronisation, graph-capturable (the clear is a kernel, the pointer is a launch argument).
    Whoever reads the results on the host anyway (submission.predict_fused, the per-pair plugin, the census) calls `rerun_out_of_range`, which runs
    the batch again through the EXACT twin of the pipeline (options SPLIT = 'bf16x3': three-term bf16 operands, the fp32 exponent range)."""

    def __init__(self, device, make_twin):
        self.active = options.get("SPLIT") == "f16x2"
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self._make_twin, self._twin, self.reruns = make_twin, None, 0

    def __enter__(self):
        if self.active:
            self.flag.zero_()
            _lib.check(_lib.load().mfr_f16x2_guard_bind(self.flag.data_ptr()), "mfr_f16x2_guard_bind")
        return self

    def __exit__(self, *exc):
        if self.active:
            _lib.load().mfr_f16x2_guard_bind(None)
        return False

    def keep(self):
        """context manager that binds the flag WITHOUT clearing it: a later stage of the same pair / batch (LoFTR's fine stage after the match count)"""
        g = self

        class _Keep:
            def __enter__(self):
                if g.active:
                    _lib.check(_lib.load().mfr_f16x2_guard_bind(g.flag.data_ptr()), "mfr_f16x2_guard_bind")

            def __exit__(self, *exc):
                return g.__exit__()
        return _Keep()

    def snapshot(self):
        """a copy of the flag, made on the CURRENT stream right after the matcher stage: what a solver stage running on another stream folds into the
        status while the next batch's matcher already clears and re-uses the flag itself"""
        return self.flag.clone() if self.active else None

    def fold(self, status, flag=None):
        flag = self.flag if flag is None else flag
        return torch.where(flag > 0, torch.full_like(status, ST_RANGE), status) if self.active else status

    def twin(self):
        if self._twin is None:
            with options.override(SPLIT="bf16x3"):
                self._twin = self._make_twin()
        return self._twin


class SolverOverlap:
    """The solver stage of batch i on a SECOND HIP stream, so that it runs under the matcher stage of batch i + 1 (opt-in: `overlap_solver=True`).

    The RANSAC kernels are latency-bound on small grids -- `pnp_select` is 32 wavefronts on a 1024-SIMD chip, `emat_roots` 16 k threads -- while the
    matcher's convolutions fill the chip: run back to back on one stream the solver's 0.7 ms (SuperGlue + PnP) / 2.9 ms (LoFTR + E-mat) per step are
    almost idle time.  Ordering: the side stream waits for an event recorded after the matcher stage; every tensor it reads that the main stream
    produced is `record_stream`ed (the caching allocator must not recycle it early); successive solver stages are ordered by the side stream itself
    (they share the solvers' workspaces).  The CONSUMER of the results must call `join()` (pipeline.join()) on the stream it reads them from --
    bench.py does so once, before the gather that ends the timed region."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.done, self._outs = None, []

    def run(self, fn, consumed):
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            out = fn()
        for t in consumed:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.stream)
        self.done = torch.cuda.Event()
        self.done.record(self.stream)
        self._outs = [t for t in out.values() if isinstance(t, torch.Tensor) and t.is_cuda]
        return out

    def join(self):
        if self.done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.done)
            for t in self._outs:                      # allocated on the side stream, read on this one from here on
                t.record_stream(cur)
            self.done, self._outs = None, []


def rerun_out_of_range(pipe, out, *args, **kw):
    """out = pipe(*args, **kw) came back: if the range guard marked the batch (status ST_RANGE; reading it synchronises with the device, so call this
    where the results are read anyway), run the batch again in the exact bf16x3 arithmetic and hand out THAT result"""
    if hasattr(pipe, "join"):
        pipe.join()
    g = getattr(pipe, "guard", None)
    if g is None or not g.active or not bool((out["status"] == ST_RANGE).any()):
        return out
    g.reruns += 1
    with options.override(SPLIT="bf16x3"):                  # layers that resolve the option on their first call (nets/loftr.py) do so in here
        return g.twin()(*args, **kw)


class SuperGluePnPPipeline:
    def __init__(self, device="cuda", sp_state=None, sg_state=None, max_keypoints=1024,
                 pnp_iters=1000, pnp_thr=3.0, pnp_conf=0.9999, seed=0, graph=False, overlap_solver=False):
        """overlap_solver=True: the PnP stage runs on a second stream under the next call's matcher (SolverOverlap; call join() before reading results).
        graph=True: the whole step (SuperPoint x2 -> SuperGlue -> lift -> PnP-RANSAC, ~330 kernel launches) is captured once
        per batch shape and replayed as ONE HIP graph; inputs are copied into the graph's static buffers on every call and the
        (small) results are handed out as copies (nets/graph.py)."""
        _lib.load(require_gpu=True)
        self.graph, self._graphs = bool(graph), {}
        self.device = torch.device(device)
        self.sp = SuperPointHIP(sp_state or WT.superpoint_state_dict(), self.device, max_keypoints=max_keypoints)
        self.sg = SuperGlueHIP(sg_state or WT.superglue_state_dict(), self.device)
        self.pnp = PnPBatchSolver(pnp_iters, pnp_thr, pnp_conf, seed)
        self.K = max_keypoints
        self.guard = RangeGuard(self.device, lambda: SuperGluePnPPipeline(device, sp_state, sg_state, max_keypoints, pnp_iters, pnp_thr, pnp_conf, seed, False))
        self._ov = SolverOverlap(self.device) if (overlap_solver and not graph) else None

    def join(self):
        """make the current stream wait for the solver stage of the last call (only needed with overlap_solver=True)"""
        if self._ov is not None:
            self._ov.join()

    @torch.no_grad()
    def match(self, images):
        """images [2B,1,H,W] -> dict(pts0, pts1 [B,K,2], n_corr [B], ...)  (SuperGlue_matcher.match)"""
        H, W = images.shape[-2:]
        sp = self.sp(images)
        m = self.sg(sp, (H, W), maxN=self.K)
        m["n_kpts"] = sp["n"]
        return m

    def __call__(self, images, depth0, K0, K1, pair_ids, want_mask=False):
        if self.graph and not want_mask:
            from .nets.graph import GraphCaptureError, GraphedCall
            args = [images, depth0, K0, K1, pair_ids]
            key = tuple((tuple(a.shape), a.dtype) for a in args)      # dtype too: float32 K and float64 K are different arithmetic (k_dtype)
            try:
                if key not in self._graphs:
                    self._graphs[key] = GraphedCall(self._run, args, clone_outputs=True)
                return self._graphs[key](*args)
            except GraphCaptureError as e:
                import warnings
                warnings.warn(f"SuperGluePnPPipeline: {e}; running eagerly from now on")
                self.graph = False
        return self._run(images, depth0, K0, K1, pair_ids, want_mask)

    @torch.no_grad()
    def _run(self, images, depth0, K0, K1, pair_ids, want_mask=False):
        with self.guard:
            m = self.match(images)
        if self._ov is not None:
            bad = self.guard.snapshot()

            def solve():
                o = self.pnp(m["pts0"], m["pts1"], m["n_corr"], depth0, K0, K1, pair_ids, want_mask=want_mask)
                o["status"] = self.guard.fold(o["status"], bad)
                return o
            out = self._ov.run(solve, [m["pts0"], m["pts1"], m["n_corr"], bad, depth0, K0, K1, pair_ids])
        else:
            out = self.pnp(m["pts0"], m["pts1"], m["n_corr"], depth0, K0, K1, pair_ids, want_mask=want_mask)
            out["status"] = self.guard.fold(out["status"])
        out["n_corr"] = m["n_corr"]
        out["pts0"], out["pts1"], out["n_kpts"] = m["pts0"], m["pts1"], m["n_kpts"]
        return out


class LoFTREmatPipeline:
    """LoFTR coarse-to-fine matching -> E-matrix RANSAC -> metric scale from depth
    (config/matching/mapfree/loftr_emat_dptkitti.yaml: PIX_THRESHOLD 2.0, SCALE_THRESHOLD 0.1,
    CONFIDENCE 0.9999), one device-resident pass over a batch of pairs.  Replaces
    LoFTR_matcher.match (matchers.py:24-59) + EssentialMatrixMetricSolver (pose_solver.py:115-172)."""

    def __init__(self, device="cuda", loftr_state=None, pix_thr=2.0, scale_thr=0.1, conf=0.9999, seed=0, pad_to=8, emat_score="magsac", overlap_solver=False):
        from .nets.loftr import LoFTRHIP
        from .solver_ops import EssentialBatchSolver, ScaleFromDepthBatch
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.loftr = LoFTRHIP(loftr_state or WT.loftr_state_dict(), self.device)
        self.emat = EssentialBatchSolver(pix_thr, conf, seed, score=emat_score)
        self.scale = ScaleFromDepthBatch(scale_thr)
        self.pad_to = pad_to
        self.guard = RangeGuard(self.device, lambda: LoFTREmatPipeline(device, loftr_state, pix_thr, scale_thr, conf, seed, pad_to, emat_score))
        self._ov = SolverOverlap(self.device) if overlap_solver else None      # E-mat + scale stage under the next call's matcher; join() before reading

    def join(self):
        if self._ov is not None:
            self._ov.join()

    @torch.no_grad()
    def match(self, images):
        H, W = images.shape[-2:]
        # LoFTR needs multiples of 8; the reference right-pads W 540 -> 544 (matchers.py:41-46, quirk Q3)
        ph, pw = (-H) % self.pad_to, (-W) % self.pad_to
        if ph or pw:
            images = torch.nn.functional.pad(images, (0, pw, 0, ph))
        return self.loftr(images)

    @torch.no_grad()
    def __call__(self, images, depth0, depth1, K0, K1, pair_ids):
        with self.guard:
            m = self.match(images)
        bad = self.guard.snapshot() if self._ov is not None else None

        def solve():
            e = self.emat(m["pts0"], m["pts1"], m["n_corr"], K0, K1, pair_ids)
            s = self.scale(m["pts0"], m["pts1"], e["mask"], m["n_corr"], depth0, depth1, K0, K1, e["R"], e["t"], e["status"])
            s["status"] = self.guard.fold(s["status"], bad)
            return dict(R=torch.where((s["status"] == 0)[:, None, None], e["R"], torch.full_like(e["R"], float("nan"))),
                        t=s["t_metric"], n_inliers=s["n_inliers"], status=s["status"], emat_inliers=e["n_inliers"], emat_mask=e["mask"])
        out = self._ov.run(solve, [m["pts0"], m["pts1"], m["n_corr"], bad, depth0, depth1, K0, K1, pair_ids]) if self._ov is not None else solve()
        out.update(n_corr=m["n_corr"], pts0=m["pts0"], pts1=m["pts1"])
        return out


# ---------------------------------------------------------------------------------------------------------------
# config-driven fused pipeline: any matcher stage x any solver stage, batched (what submission.predict_fused runs)
# ---------------------------------------------------------------------------------------------------------------
class _PrecomputedStage:
    """cfg.FEATURE_MATCHING == 'Precomputed' (feature_matching.py:5-50) for a batch: npz rows -> device layout"""

    def __init__(self, cfg, device):
        from .matching.feature_matching import PrecomputedMatching
        self.pm, self.device = PrecomputedMatching(cfg), device

    def __call__(self, batch):
        import numpy as np
        from . import wire
        rows = []
        pids = batch["seed_ids"].tolist()
        sids = batch.get("scene_ids") or [batch["scene_id"]] * len(pids)           # batches may span scenes (PairBatchLoader)
        roots = batch.get("scene_roots") or [batch.get("scene_root", "")] * len(pids)
        for pid, sid, root in zip(pids, sids, roots):
            p1, p2 = self.pm.get_correspondences({"scene_id": [sid], "scene_root": [root], "pair_id": pid})
            rows.append(np.concatenate([p1, p2], 1) if len(p1) else np.zeros((0, 4), np.float32))
        p0, p1, n = wire.pts_rows_to_device_batch(rows)
        d = lambda a: torch.from_numpy(a).to(self.device)
        return dict(pts0=d(p0), pts1=d(p1), n_corr=d(n))


class _RefCachedSuperGlue:
    """SuperPoint + SuperGlue over a batch of pairs with ONE SuperPoint pass per distinct reference view.

    Every val / test pair of a Map-free scene has the same reference image, seq0/frame_00000.jpg
    (etc/feature_matching_baselines/compute.py:72-75, lib/datasets/mapfree.py:148-165); the reference runs the detector on it once per
    pair (116 times per scene).  Here the batch's pairs are grouped by `ref_keys` (PairBatchLoader: (scene_root, reference frame
    name)); SuperPoint runs on [the reference views not seen yet | all query views]; the reference rows of the interleaved
    [2B] keypoint / score / descriptor tensors SuperGlue consumes are gathered from a small cache (the last few reference views: a
    scene's pairs span several batches).  Every SuperPoint stage is per image with a fixed summation order (own convolutions, own 1x1
    heads), so the result is the SAME BITS as the plain path -- tests/test_gpu_fused_submission.py checks poses byte for byte."""

    def __init__(self, sp, net, max_kpts, plain, keep=4):
        import collections
        self.sp, self.net, self.K, self.plain, self.keep = sp, net, max_kpts, plain, keep
        self.cache = collections.OrderedDict()
        self.stats = dict(reference_views_run=0, reference_views_reused=0)

    def __call__(self, b):
        im, keys = b["images"], b.get("ref_keys")
        if keys is None or len(keys) * 2 != im.shape[0]:
            return self.plain(im)
        B = len(keys)
        keys = [k if k is not None else ("__pair__", p) for p, k in enumerate(keys)]      # None: a reference view nobody shares
        first = {}
        for p, k in enumerate(keys):
            first.setdefault(k, p)
        need = [k for k in first if k not in self.cache]
        parts = [im[2 * first[k]][None] for k in need] + [im[1::2]]
        out = self.sp(torch.cat(parts) if need else im[1::2].contiguous())
        u = len(need)
        for i, k in enumerate(need):
            self.cache[k] = {f: out[f][i:i + 1].clone() for f in ("kpts", "scores", "desc", "n")}
        self.stats["reference_views_run"] += u
        self.stats["reference_views_reused"] += B - u
        order = list(first)
        sp2 = {}
        for f in ("kpts", "scores", "desc", "n"):
            ref = torch.cat([self.cache[k][f] for k in keys])                      # [B, ...] reference rows (no index tensor: an H2D copy would
            qry = out[f][u:]                                                       # synchronise the stream once per batch)
            sp2[f] = torch.stack([ref, qry], 1).reshape((2 * B,) + tuple(qry.shape[1:]))
        for k in order:                                                           # most recently used last; bounded
            if k[0] == "__pair__":
                del self.cache[k]
            else:
                self.cache.move_to_end(k)
        while len(self.cache) > max(self.keep, len(order)):
            self.cache.popitem(last=False)
        return self.net(sp2, tuple(im.shape[-2:]), maxN=self.K)


class FusedPosePipeline:
    """The batched twin of FeatureMatchingModel (lib/models/matching/model.py:7-40): the same two config keys pick the
    stages -- FEATURE_MATCHING in {'Precomputed', 'SuperGlue', 'LoFTR'} x POSE_SOLVER in {'PNP', 'EssentialMatrix',
    'EssentialMatrixMetric', 'Procrustes'} -- but every stage consumes / produces a device-resident batch of pairs.
    __call__(batch) with the dict PairBatchLoader yields -> dict(R [b,3,3] f64, t [b,3] f64, n_inliers, status, n_corr)."""

    def __init__(self, cfg, device="cuda"):
        from . import solver_ops as ops
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.guard = RangeGuard(self.device, lambda: FusedPosePipeline(cfg, device))
        seed = int(cfg.RANSAC.SEED) if "RANSAC" in cfg else 0
        fm = cfg.FEATURE_MATCHING
        if fm == "Precomputed":
            self.match = _PrecomputedStage(cfg, self.device)
        elif fm == "SuperGlue":
            sg = cfg.SUPERGLUE
            sp_sd = WT.load_checkpoint(sg.SUPERPOINT_WEIGHTS) if sg.SUPERPOINT_WEIGHTS else WT.synthetic_or_raise("SuperPoint", cfg, WT.superpoint_state_dict)
            sg_sd = WT.load_checkpoint(sg.SUPERGLUE_WEIGHTS) if sg.SUPERGLUE_WEIGHTS else WT.synthetic_or_raise("SuperGlue", cfg, WT.superglue_state_dict)
            sp = SuperPointHIP(sp_sd, self.device, sg.NMS_RADIUS, sg.KEYPOINT_THRESHOLD, sg.MAX_KEYPOINTS)
            net = SuperGlueHIP(sg_sd, self.device, sg.SINKHORN_ITERATIONS, sg.MATCH_THRESHOLD)
            fwd = lambda images: net(sp(images), tuple(images.shape[-2:]), maxN=sg.MAX_KEYPOINTS)
            if "GRAPH_FUSED" in cfg.HIP and cfg.HIP.GRAPH_FUSED:
                # the matcher stage (~300 launches) replayed from one HIP graph per batch shape; at most two shapes are captured
                # (the full batch and one remainder), anything else runs eagerly: every capture pins its intermediates
                from .nets.graph import GraphCaptureError, GraphedCall
                graphs = {}

                def match(b):
                    im = b["images"]
                    key = tuple(im.shape)
                    if key not in graphs:
                        if len(graphs) >= 2 or not im.is_cuda or graphs.get("failed"):
                            return fwd(im)
                        try:
                            graphs[key] = GraphedCall(torch.no_grad()(fwd), [im], clone_outputs=True)
                        except GraphCaptureError as e:
                            import warnings
                            warnings.warn(f"FusedPosePipeline: {e}; the matcher stage runs eagerly")
                            graphs["failed"] = True
                            return fwd(im)
                    return graphs[key](im)
                self.match = match
            elif "REF_FEATURE_CACHE" in cfg.HIP and cfg.HIP.REF_FEATURE_CACHE:
                self.match = _RefCachedSuperGlue(sp, net, sg.MAX_KEYPOINTS, fwd)
            else:
                self.match = lambda b: fwd(b["images"])
        elif fm == "LoFTR":
            lw = cfg.LOFTR.WEIGHTS
            sd = WT.strip_prefix(WT.load_checkpoint(lw), "matcher.") if lw else WT.synthetic_or_raise("LoFTR", cfg, WT.loftr_state_dict)
            lp = LoFTREmatPipeline.__new__(LoFTREmatPipeline)
            from .nets.loftr import LoFTRHIP
            lp.loftr, lp.pad_to, lp.device = LoFTRHIP(sd, self.device), 8, self.device
            self.match = lambda b: lp.match(b["images"])
        else:
            raise NotImplementedError(f"FEATURE_MATCHING={fm!r} has no batched stage (SIFT detection is OpenCV / per pair)")
        ps = cfg.POSE_SOLVER
        if ps == "PNP":
            pnp = ops.PnPBatchSolver(cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE, seed)
            self.solve = lambda m, b: pnp(m["pts0"], m["pts1"], m["n_corr"], b["depth0"], b["K0"], b["K1"], b["seed_ids"])
        elif ps in ("EssentialMatrix", "EssentialMatrixMetric"):
            em = ops.EssentialBatchSolver(cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE, seed,
                                          score=cfg.HIP.EMAT_SCORE, max_thr_ratio=cfg.HIP.MAGSAC_MAX_THR_RATIO)
            sc = ops.ScaleFromDepthBatch(cfg.EMAT_RANSAC.SCALE_THRESHOLD) if ps == "EssentialMatrixMetric" else None

            def solve(m, b):
                e = em(m["pts0"], m["pts1"], m["n_corr"], b["K0"], b["K1"], b["seed_ids"])
                if sc is None:
                    return e
                s = sc(m["pts0"], m["pts1"], e["mask"], m["n_corr"], b["depth0"], b["depth1"], b["K0"], b["K1"], e["R"], e["t"], e["status"])
                return dict(R=e["R"], t=s["t_metric"], n_inliers=s["n_inliers"], status=s["status"])
            self.solve = solve
        elif ps == "Procrustes":
            pr = ops.ProcrustesBatchSolver(cfg.PROCRUSTES.MAX_CORR_DIST, 0.999, seed)
            icp = ops.ProcrustesIcpRefine(cfg.PROCRUSTES.MAX_CORR_DIST, 1e-4, 1e-4, 30) if cfg.PROCRUSTES.REFINE else None

            def solve_pr(m, b):
                o = pr(m["pts0"], m["pts1"], m["n_corr"], b["depth0"], b["depth1"], b["K0"], b["K1"], b["seed_ids"])
                if icp is not None:
                    o = dict(o, n_inliers=icp(b["depth0"], b["depth1"], b["K0"], b["K1"], o["R"], o["t"], o["status"])["n_inliers"])
                return o
            self.solve = solve_pr
        else:
            raise NotImplementedError(f"POSE_SOLVER={ps!r}")

    @torch.no_grad()
    def __call__(self, batch):
        with self.guard:
            m = self.match(batch)
        out = self.solve(m, batch)
        out["status"] = self.guard.fold(out["status"])
        ok = (out["status"] == 0)
        nan = float("nan")
        return dict(R=torch.where(ok[:, None, None], out["R"], torch.full_like(out["R"], nan)),
                    t=torch.where(ok[:, None], out["t"], torch.full_like(out["t"], nan)),
                    n_inliers=torch.where(ok, out["n_inliers"], torch.zeros_like(out["n_inliers"])), status=out["status"],
                    n_corr=m["n_corr"])
