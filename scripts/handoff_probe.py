#!/usr/bin/env python3
"""Milestone 1 of the layer-pipelined training backward: sweep of mipnerf_handoff_probe -- 128 producer
workgroups streaming 256-sample x 512 B tiles to 128 consumer workgroups through rings in global memory with counter flags --
over placement (same XCD / neighbouring XCD), store flavour (plain + agent release / sc1 write-through), ring depth, tile size
and the amount of MFMA work per tile on both sides.  Prints one JSON line per configuration (gpurun_out/r03_handoff_probe.jsonl).

    python scripts/handoff_probe.py [--out gpurun_out/r03_handoff_probe.jsonl]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r03_handoff_probe.jsonl")
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only-waves", action="store_true", help="only the per-wave protocols (flavours 3 / 4)")
    args = ap.parse_args()
    import torch
    from mipnerf_pl_amd import _lib as L
    torch.cuda.init()
    st = torch.cuda.current_stream().cuda_stream
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    rows = []
    grid = []
    for tile_bytes in ((131072,) if args.quick else (131072, 65536, 262144)):
        for same in (1, 0):
            for flavour in (0, 1, 2, 3, 4):
                for ring in ((4,) if args.quick else (2, 4, 8)):
                    for mfma in (0, 256, 512):
                        if tile_bytes != 131072 and (ring != 4 or mfma == 512):
                            continue
                        if (flavour in (2, 4) and not same) or (flavour >= 3 and tile_bytes > 131072) or (args.only_waves and flavour < 3):
                            continue
                        grid.append((same, flavour, ring, tile_bytes, mfma))
    with open(args.out, "w") as f:
        for same, flavour, ring, tile_bytes, mfma in grid:
            out = (C.c_double * 6)()
            rc = L.diag_lib().mipnerf_handoff_probe(same, flavour, args.tiles, ring, tile_bytes, mfma, 3, out, st)
            row = {"same_xcd": same, "stores": ("plain + agent release", "sc1 write-through", "plain, no fence; consumer loads bypass L1 (sc1)",
                                                                   "per-wave streams: sc1 write-through, per-wave acquire", "per-wave streams: plain, no fence, sc1 loads")[flavour], "ring": ring,
                   "tile_bytes": tile_bytes, "tiles_per_pair": args.tiles, "mfma_per_wave_per_tile": mfma, "rc": rc,
                   "msg": (L.diag_lib().mipnerf_diag_last_error() or b"").decode()}
            if rc == 0:
                # the MFMA filler alone bounds the rate: tiles * mfma * 8 waves * 32 cycles ... reported as the time the same loop
                # would take with no hand-off at 2.0 GHz is left to the reader; stall fractions say who waited
                row.update({"aggregate_GBps": round(out[0], 1), "ms": round(out[1], 4), "producer_stall_frac": round(out[2], 4),
                            "consumer_stall_frac": round(out[3], 4), "bad_words": int(out[4]), "timed_out": int(out[5]),
                            "active_pairs": int(os.environ.get("MIPNERF_PROBE_PAIRS", "128")),
                            "per_pair_GBps": round(out[0] / int(os.environ.get("MIPNERF_PROBE_PAIRS", "128")), 2),
                            "samples_per_s_at_512B": round(out[0] * 1e9 / 512, 1)})
            rows.append(row)
            line = json.dumps(row)
            print(line, flush=True)
            f.write(line + "\n")
    ok = [r for r in rows if r.get("rc") == 0 and not r["timed_out"] and not r["bad_words"]]
    if ok:
        best = max(ok, key=lambda r: r["aggregate_GBps"])
        print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
