"""MFMA study of the shared-factor KKT solve (north_star: "MFMA only if a dense-block KKT variant proves bandwidth-bound").
The shared-factor kernel of config 2 is LDS-bandwidth bound; because the factor is SHARED by the instances, 16 instances can
be the N dimension of v_mfma_f64_16x16x4_f64, with 16 x 4 tiles of a phase's matrix as the A operand and the work vectors of
the 16 instances ([slot][instance] in LDS) as the B operand.  This script cuts MPC 12/4/10's 13-phase solve program
(solve_program.compile_ldl, the program the kernel runs) into such tiles and writes
    out/mfma_program.bin   header | tile descriptors | dense tiles | row-block descriptors | reference solution of 16 right-hand sides
for scripts/micro/mfma_shared.hip, and prints the tile statistics (fill ratio, LDS / L2 bytes per instance-solve).
    python scripts/micro/mfma_gen.py [out/mfma_program.bin]"""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from cvxpygen_amd import families, solve_program as spm
from cvxpygen_amd.runtime import build_family_plan

TR, TC, NI = 16, 4, 16
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'out', 'mfma_program.bin')
d = families.mpc(12, 4, 10)
plan = build_family_plan(d)
N = d.n_var + d.m
sh = plan.osqp_shared
devpos = np.concatenate([plan.posx, d.n_var + plan.posz])
phases = spm.compile_ldl(N, sh.Lp, sh.Li, sh.Lx, sh.D, sh.perm, merge=True, devpos=devpos)
outs, ins, n_slots, fpos = spm.assign_slots(phases, N)
tiles_desc, tiles_val, rb_desc, phase_rb = [], [], [], []
tot_nnz = 0
stats = []
for pi, (ph, o, i) in enumerate(zip(phases, outs, ins)):
    R = len(ph.rows)
    rows = np.concatenate([np.full(len(c), r) for r, c in enumerate(i)])
    cols = np.concatenate([np.asarray(c) for c in i])
    vals = np.concatenate([np.asarray(v, dtype=np.float64) for v in ph.vals])
    M = sp.csr_matrix((vals, (rows, cols)), shape=(R, n_slots))
    tot_nnz += M.nnz
    # rows ordered by their first column, the used columns by the first (reordered) row that reads them: neighbours share tiles
    fr = np.array([M[r].indices.min() if M[r].nnz else 0 for r in range(R)])
    rord = np.argsort(fr, kind='stable')
    Mc = sp.csc_matrix(M[rord])
    used = np.nonzero(np.diff(Mc.indptr))[0]
    fc = np.array([Mc[:, c].indices.min() for c in used])
    cord = used[np.argsort(fc, kind='stable')]
    Md = Mc[:, cord].toarray()
    nrb, ncb = -(-R // TR), -(-len(cord) // TC)
    Md = np.pad(Md, ((0, nrb * TR - R), (0, ncb * TC - len(cord))))
    slot_of_col = np.concatenate([cord, np.full(ncb * TC - len(cord), n_slots)])       # padding columns read the zero slot
    out_slot = np.concatenate([np.asarray(o)[rord], np.full(nrb * TR - R, n_slots + 1)])   # padding rows write a dummy slot
    first_rb = len(rb_desc)
    nt = 0
    for rb in range(nrb):
        t0 = len(tiles_desc)
        for cb in range(ncb):
            blk = Md[rb * TR:(rb + 1) * TR, cb * TC:(cb + 1) * TC]
            if np.any(blk != 0.0):
                tiles_desc.append(slot_of_col[cb * TC:(cb + 1) * TC].astype(np.uint16))
                tiles_val.append(blk.T.reshape(-1).copy())          # A[i][k] at lane i + 16 k
        rb_desc.append((t0, len(tiles_desc) - t0, out_slot[rb * TR:(rb + 1) * TR].astype(np.uint16)))
        nt += len(tiles_desc) - t0
    phase_rb.append((first_rb, nrb))
    stats.append((pi, ph.name, R, len(cord), M.nnz, nt, M.nnz / (64.0 * max(nt, 1))))
T = len(tiles_desc)
print(f'MPC 12/4/10 shared solve program: {len(phases)} phases, {tot_nnz} coefficients, {n_slots} slots')
for s in stats:
    print('  phase %2d %-8s rows %4d columns %4d nnz %5d tiles %4d fill %.3f' % s)
print(f'tiles of {TR} x {TC}: {T}, fill {tot_nnz / (64.0 * T):.3f}; dense tiles {T * 512 / 1024:.0f} KiB (the sparse program: {tot_nnz * 10 / 1024:.0f} KiB of coefficients + offsets)')
print(f'per solve of {NI} instances: {T} MFMA (16x16x4 f64: 2048 flop, {T * 2048 / (2.0 * tot_nnz * NI):.1f} x the useful flops), A operand {T * 512 / 1024:.0f} KiB, '
      f'B operand {T * 512 / 1024:.0f} KiB of LDS reads; work vectors of {NI} instances {(n_slots + 2) * NI * 8 / 1024:.0f} KiB of LDS')
print(f'per INSTANCE-solve: {T * 1024 / NI / 1024:.1f} KiB of operand bytes (today: {tot_nnz * 18 / 1024:.1f} KiB of LDS reads: coefficient + offset + operand per non-zero)')
# reference: the phases applied to 16 random right-hand sides
rng = np.random.default_rng(3)
W = np.zeros((n_slots + 2, NI)); W[:N] = rng.standard_normal((N, NI))
W0 = W.copy()
for ph, o, i in zip(phases, outs, ins):
    new = np.stack([np.asarray(v) @ W[np.asarray(c)] for c, v in zip(i, ph.vals)]) if len(ph.rows) else np.zeros((0, NI))
    W[np.asarray(o)] = new
with open(out, 'wb') as f:
    f.write(struct.pack('8i', len(phases), T, len(rb_desc), n_slots + 2, NI, 0, 0, 0))
    f.write(np.asarray([v for p in phase_rb for v in p], dtype=np.int32).tobytes())
    f.write(np.concatenate(tiles_desc).astype(np.uint16).tobytes())
    f.write(np.concatenate(tiles_val).astype(np.float64).tobytes())
    f.write(np.asarray([(t0, n) for t0, n, _ in rb_desc], dtype=np.int32).tobytes())
    f.write(np.concatenate([o_ for _, _, o_ in rb_desc]).astype(np.uint16).tobytes())
    f.write(W0.astype(np.float64).tobytes())
    f.write(W.astype(np.float64).tobytes())
print('wrote', out, os.path.getsize(out), 'bytes')
