"""bank-conflict check of ds_read_b128 fragment reads on 128-byte LDS rows (gfx950 lane groups)"""
import itertools
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
def worst(f, shift, h):
    w = 0
    for grp in groups:
        banks = {}
        for lane in grp:
            l15, g = lane & 15, lane >> 4
            r = l15 + shift
            c = (g + 4*h) ^ f(r)
            addr = r*128 + c*16
            for d in range(4):
                b = (addr//4 + d) % 64
                banks.setdefault(b, set()).add(addr//4 + d)
        w = max(w, max(len(v) for v in banks.values()))
    return w
cands = {
 'r&7': lambda r: r & 7,
 '(r>>1)&7': lambda r: (r >> 1) & 7,
 'r&6': lambda r: r & 6,
 '(r>>1)&3': lambda r: (r >> 1) & 3,
 '((r>>1)&3)<<1': lambda r: ((r >> 1) & 3) << 1,
 '(r>>1)&1': lambda r: (r >> 1) & 1,
 '((r>>1)&1)<<2': lambda r: ((r >> 1) & 1) << 2,
 'none': lambda r: 0,
}
for name, f in cands.items():
    res = [[worst(f, s, h) for s in range(0, 11)] for h in (0, 1)]
    print(f"{name:16s}", res)
