#!/bin/bash
# round 3: the whole hand-off sweep again (workgroup-wide protocols 0-2 and the per-wave protocols 3 / 4 of probe v2), then the
# per-wave protocols with fewer active pairs (is ~20 GB/s per pair a per-CU or a chip-wide limit?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/handoff_probe.py --out gpurun_out/r03w_handoff_all.jsonl > /dev/null 2>&1
for p in 64 32 16 8; do
  MIPNERF_PROBE_PAIRS=$p timeout 300 python scripts/handoff_probe.py --only-waves --quick --out gpurun_out/r03w_handoff_waves_pairs$p.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03w_handoff_*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        if d["ring"] != 4: continue
        print(f.split("/")[-1][13:-6].ljust(14), d["same_xcd"], d["stores"][:40].ljust(40), d["tile_bytes"] // 1024, "mfma", d["mfma_per_wave_per_tile"], "GB/s", d.get("aggregate_GBps"), "per pair", d.get("per_pair_GBps"), "bad", d.get("bad_words"), d.get("timed_out"))
PY
