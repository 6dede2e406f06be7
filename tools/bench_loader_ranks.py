"""The host side of predict_fused alone, as N ranks of one host run it: every rank walks its contiguous share of the scenes with its own
PairBatchLoader (decode = thread | process, workers = the granted CPUs / N like submission.predict_fused), nothing consumes the batches but a
counter.  Reports pairs/s per rank and in total -- the rate the loaders can feed N GPUs at, and what sharing the host's cores between ranks
costs ("state the per-rank stall at LOCAL_WORLD_SIZE 2").  No GPU needed.

python tools/bench_loader_ranks.py --root /tmp/mapfree_loader --scenes 8 --frames 24 --ranks 1,2 --decode thread,process [--out profiles/...json]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def rank_main(a):
    root, rank, world, decode, B, q, barrier = a
    import mapfree_reloc_amd  # noqa: F401
    from mapfree_reloc_amd import parallel
    from mapfree_reloc_amd.config import get_cfg_defaults
    from mapfree_reloc_amd.datasets import PairBatchLoader, list_scenes, usable_cpus
    cfg = get_cfg_defaults()
    cfg.DATASET.DATA_ROOT = root
    scenes = list_scenes(cfg, "test")
    lo, hi = parallel.shard_scenes([len(sc) for sc in scenes], world)[rank]
    workers = max(2, min(32, usable_cpus() // world))
    loader = PairBatchLoader(scenes[lo:hi], B, prefetch=2, pin=False, workers=workers, decode=decode)
    barrier.wait()
    t0 = time.perf_counter(); n = 0
    for b in loader:
        n += len(b["names"])
    dt = time.perf_counter() - t0
    loader.close()
    q.put(dict(rank=rank, pairs=n, seconds=round(dt, 3), pairs_per_s=round(n / dt, 1), workers=workers))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="/tmp/mapfree_loader"); ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--ranks", default="1,2"); ap.add_argument("--decode", default="thread,process"); ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from bench_fused_split import write_scene
    if not os.path.isdir(os.path.join(a.root, "test", f"s{a.scenes - 1:05d}")):
        for s in range(a.scenes):
            write_scene((a.root, s, a.frames))
    from mapfree_reloc_amd.datasets import usable_cpus
    res = {"host": {"visible_cpus": os.cpu_count(), "granted_cpus": usable_cpus()}, "tree": {"scenes": a.scenes, "pairs": a.scenes * a.frames}, "runs": []}
    ctx = mp.get_context("spawn")
    for decode in a.decode.split(","):
        for world in [int(w) for w in a.ranks.split(",")]:
            q, barrier = ctx.Queue(), ctx.Barrier(world)
            ps = [ctx.Process(target=rank_main, args=((a.root, r, world, decode, a.batch, q, barrier),)) for r in range(world)]
            for p in ps: p.start()
            rows = sorted([q.get() for _ in ps], key=lambda r: r["rank"])
            for p in ps: p.join()
            wall = max(r["seconds"] for r in rows)
            rec = dict(decode=decode, ranks=world, per_rank=rows, total_pairs_per_s=round(sum(r["pairs"] for r in rows) / wall, 1))
            res["runs"].append(rec)
            print(json.dumps(rec), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
