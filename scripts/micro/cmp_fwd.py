import numpy as np, sys
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
print("raw equal", np.array_equal(a["raw"], b["raw"]))
ma, mb = a["masks"].reshape(-1, 9, 1024), b["masks"].reshape(-1, 9, 1024)
for l in range(9):
    print("mask row", l, "diff bytes", int((ma[:, l] != mb[:, l]).sum()))
aa, ab = a["act"].reshape(ma.shape[0], -1, 2048), b["act"].reshape(ma.shape[0], -1, 2048)
bad = [(blk, int((aa[:, blk] != ab[:, blk]).sum())) for blk in range(aa.shape[1]) if (aa[:, blk] != ab[:, blk]).any()]
print("act blocks differing:", bad)
wa = a["masks"].view(np.uint32).reshape(-1, 9, 64, 4)[:, 8]
wb = b["masks"].view(np.uint32).reshape(-1, 9, 64, 4)[:, 8]
for d in range(4):
    print("dword", d, "lanes differing", int((wa[..., d] != wb[..., d]).sum()), "of", wa[..., d].size)
idx = np.argwhere(wa != wb)[:6]
for (w, l, d) in idx:
    print(f"wave {w} lane {l} dword {d}: good {wa[w, l, d]:08x} new {wb[w, l, d]:08x} xor {wa[w, l, d] ^ wb[w, l, d]:08x}")
