#!/usr/bin/env python3
"""Is there a LINEAR row stride that makes the ring kernel's transposed LDS reads conflict-free?  (No: hence the XOR of the 32-byte piece.)

gemm_nt3r_kernel (csrc/wgrad_nt3r.hip) reads its MFMA B operand with ds_read_b64_tr_b16: per 32-lane half, 8 row pieces of 32 bytes of one 16-channel
block.  The dH side fixes which rows: a window's 8 time steps are loaded as two 4-step tuples and split in place, dword q = (tuple 1 step q, tuple 2
step q), so a read's four rows are (T1 + e, T2 + e, T1 + e + 1, T2 + e + 1) for e = 0 (first read) or 2 (second), with tuple starts T multiples of 4.
The bank of byte a is (a / 4) % 64 for this instruction (cdna_hip_programming.md section 2), i.e. the 8 pieces must land on 8 distinct 32-byte groups
of a 256-byte bank row.  Brute force over the row stride (in 8-byte units) and over every assignment of the 8 tuples of a 32-step k-step to the two
halves: prints the strides that work -- none, for any stride from 32 to 512 bytes."""
import itertools


def ok(rows, k):
    occ = set()
    for r in rows:
        for u in range(4):
            b = (r * k + u) % 32
            if b in occ:
                return False
            occ.add(b)
    return True


found = []
for k in range(4, 65):
    for sub in itertools.combinations(range(8), 4):
        comp = [i for i in range(8) if i not in sub]
        if all(ok([4 * i + e for i in S for e in (e0, e0 + 1)], k) for S in (sub, comp) for e0 in (0, 2)):
            found.append((8 * k, sub))
            break
print("conflict-free linear strides (bytes):", found if found else "none")
