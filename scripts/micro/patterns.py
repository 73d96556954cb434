"""patterns for lds_conflicts: 64 slot numbers per line; prints the model cost next to each (stderr)"""
import sys
import numpy as np
rng = np.random.default_rng(0)
pats = []
def model(p, group=32, banks=32):
    c = 0
    for g in range(0, 64, group):
        sl = np.unique(p[g:g + group])
        c += np.bincount(sl % banks, minlength=banks).max()
    return c
L = np.arange(64)
pats.append(('contiguous', L.copy()))
pats.append(('broadcast', np.zeros(64, dtype=int)))
pats.append(('stride2', L * 2))
pats.append(('stride4', L * 4))
pats.append(('stride8', L * 8))
pats.append(('stride16', L * 16))
pats.append(('stride32', L * 32))
pats.append(('lane+32*(lane%2)', L + 32 * (L % 2)))
pats.append(('pairs l,l+16 same bank', (L % 16) + 32 * ((L // 16) % 2)))
pats.append(('pairs l,l+32 same bank', (L % 32) + 32 * (L // 32)))
pats.append(('half contiguous, half one slot', np.where(L < 32, L, 7)))
pats.append(('lane%16 + 16*(lane//16)*3 (distinct mod 16 per 16 lanes)', (L % 16) + 48 * (L // 16)))
pats.append(('pairs l,l+8 same mod16', (L % 8) + 16 * ((L // 8) % 2) + 64 * (L // 16)))
for k in range(12):
    pats.append((f'random{k}', rng.integers(0, 1024, 64)))
for k in range(6):
    pats.append((f'random16_{k}', rng.integers(0, 64, 64)))
for name, p in pats:
    print(' '.join(str(int(v)) for v in p))
    print(f'{name:34s} model32x32 {model(p)} model16x32 {model(p,16,32)} model32x16 {model(p,32,16)} model64x32 {model(p,64,32)} w16x16 {model(p,16,16)} w16x32 {model(p,16,32)}', file=sys.stderr)
