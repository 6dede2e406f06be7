"""Generate tests/golden/ref_submission_format.npz by EXECUTING THE REFERENCE'S OWN submission.py (build container only): the `Pose`
line formatter (submission.py:18-30) and `save_submission` (:61-65) on seeded poses -- float32 / float64 quaternions and translations,
negative zero, large and tiny magnitudes, float / int confidences.  The module's other imports (yacs config, Lightning datamodule, model
builder, transforms3d) are stubbed: none of them is touched by the two symbols executed here."""
import io
import os
import sys
import types
import zipfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub(name)

    def __call__(self, *a, **k):
        return None


def cases():
    rng = np.random.default_rng(77)
    out = []
    for i in range(24):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 10.0 ** rng.integers(-4, 3)
        if i % 5 == 0:
            t[0] = -0.0
        if i % 7 == 0:
            q, t = q.astype(np.float32), t.astype(np.float32)
        conf = [457, 12.0, 0, 3.5e-07, np.float64(17.25), np.int64(9)][i % 6]
        out.append((f"seq1/frame_{5 * i:05d}.jpg", q, t, conf))
    return out


def main():
    for name in ("config", "config.default", "lib", "lib.datasets", "lib.datasets.datamodules", "lib.models", "lib.models.builder", "lib.utils",
                 "lib.utils.data", "transforms3d", "transforms3d.quaternions", "tqdm"):
        sys.modules[name] = _Stub(name)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_submission", os.path.join(REF, "submission.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cs = cases()
    lines = [str(ref.Pose(n, q, t, c)) for n, q, t, c in cs]
    res = {"s00460": [ref.Pose(*c) for c in cs[:10]], "s00461": [], "s00462": [ref.Pose(*c) for c in cs[10:]]}
    path = os.path.join(OUT, "_tmp_ref_submission.zip")
    ref.save_submission(res, path)
    with zipfile.ZipFile(path) as zf:
        members = zf.namelist()
        texts = [zf.read(m).decode("utf-8") for m in members]
    os.remove(path)
    np.savez_compressed(os.path.join(OUT, "ref_submission_format.npz"), lines=np.array(lines), members=np.array(members), texts=np.array(texts))
    print(len(lines), members, lines[0], lines[5], sep="\n")


if __name__ == "__main__":
    main()
