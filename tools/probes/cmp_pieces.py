import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2]); c1 = int(sys.argv[3]) if len(sys.argv) > 3 else 64
MB1, MB2, NP = c1 // 32, 4, 32
blk = NP * 256
names = [("h1", MB2), ("dz1", MB2), ("h0", MB1), ("dz0", MB1), ("xb", 1)]
off = 0
for n, mbs in names:
    for mb in range(mbs):
        x, y = a[:, off:off + blk], b[:, off:off + blk]
        bad = np.argwhere(x != y)
        print(f"{n}[{mb}]: {len(bad)} of {x.size} differ", "" if not len(bad) else f"first {bad[:6].tolist()}  (octet, lane, slot) of first: {divmod(int(bad[0][1]), 256)[0], divmod(int(bad[0][1]) % 256, 4)}")
        off += blk
ga, gb = np.load(sys.argv[1].replace(".npy", "_g.npy")), np.load(sys.argv[2].replace(".npy", "_g.npy"))
print("grad max abs diff", np.abs(ga - gb).max(), "of", np.abs(ga).max())
off = 0
for n, mbs in names:
    for mb in range(mbs):
        x, y = a[:, off:off + blk], b[:, off:off + blk]
        bad = np.argwhere(x != y)
        for bb, e in bad[:40]:
            o, rem = divmod(int(e), 256); lane, slot = divmod(rem, 4)
            print(f"  {n}[{mb}] cloud {bb} octet {o} lane {lane} slot {slot}: A {x[bb, e]:.6g}  B {y[bb, e]:.6g}")
        off += blk
