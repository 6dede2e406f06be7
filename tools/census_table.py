"""Markdown parity table of DESIGN.md section 2, generated from the committed census records (tools/parity_census.py on an MI355X) so that the text and
the records cannot disagree.   python tools/census_table.py [profiles/r06_parity_census_easy.json ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or [os.path.join(ROOT, "profiles", f"r06_parity_census_{t}.json") for t in ("easy", "hard", "hard2")]
NAMES = {"sg_pnp": "configs[1] SuperPoint+SuperGlue + PnP", "loftr_emat": "configs[2] LoFTR + E-mat (MAGSAC++) + scale", "sg_procrustes": "f-1 SuperGlue + Procrustes RANSAC",
         "sift_emat": "configs[0] descriptor leg + E-mat"}
print("| config | scenes | pairs | status agree | pose within 1e-4 rad / 1e-4 m | max Δ rot (rad) / trans (m) | match set identical (same order / as a set) | "
      "inlier index sets identical (raw / at 1/64 px; min Jaccard at 1/64 px) | inlier counts equal | pose bit-equal | median inlier fraction |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for f in files:
    d = json.load(open(f))
    tag = os.path.basename(f).replace("r06_parity_census_", "").replace(".json", "")
    for k, nm in NAMES.items():
        if k not in d:
            continue
        s = d[k]["summary"]
        ins = (f'{s.get("inlier_index_sets_identical", "—")} / {s.get("inlier_index_sets_identical_q64", "—")}; {s.get("min_inlier_set_jaccard_q64", "—")}'
               if "inlier_index_sets_compared" in s else "—")
        print(f'| {nm} | {tag} | {s["pairs"]} | {s["status_agree"]} | **{s.get("pose_within_bar", "—")}** | {s.get("max_rot_rad", 0):.1e} / {s.get("max_trans_m", 0):.1e} | '
              f'{s["identical_match_sets"]} / {s["identical_as_sets"]} | {ins} | {s["inlier_count_equal"]} | {s.get("pose_bit_equal", "—")} | {s.get("median_inlier_fraction", "—")} |')
