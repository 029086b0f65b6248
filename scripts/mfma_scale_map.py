#!/usr/bin/env python3
"""Find the operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 by one-hot experiments (e4m3
operands): which (lane, byte) of B multiplies a given (lane, byte) of A, where the product lands,
and which products a lane's scale byte scales.  Prints a summary; the GPU test pins the result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kvmini.ops import native

C = native()
dev = torch.device("cuda:0")
ONE = 0x38  # e4m3 1.0
unit = torch.full((64,), 127, dtype=torch.int32, device=dev)


def regs(mat):  # uint8 [64, 32] -> int32 [64, 8]
    return mat.contiguous().view(torch.int32).view(64, 8).to(dev)


def run(a, b, sa=unit, sb=unit, fa=0, fb=0):
    return C.mfma_scale_probe(regs(a), regs(b), sa, sb, fa, fb).cpu()


ones = torch.full((64, 32), ONE, dtype=torch.uint8)
zeros = torch.zeros((64, 32), dtype=torch.uint8)
print("all ones:", run(ones, ones).unique().tolist(), "(expect 128)")
# where does row of A land / which column does B's lane give
for la in (0, 1, 15, 16, 33, 63):
    a = zeros.clone(); a[la, 0] = ONE
    out = run(a, ones)
    nz = out.nonzero().tolist()
    print(f"A one-hot lane {la} byte 0 x B ones -> nonzero (lane, reg): {sorted(set((l, e) for l, e in nz))[:20]} n={len(nz)}")
for lb in (0, 1, 15, 16, 33, 63):
    b = zeros.clone(); b[lb, 0] = ONE
    out = run(ones, b)
    nz = out.nonzero().tolist()
    print(f"B one-hot lane {lb} byte 0 x A ones -> nonzero (lane, reg): {sorted(set((l, e) for l, e in nz))[:20]} n={len(nz)}")
# pairing: A one-hot (la, ja) against B one-hot (lb, jb) in the same column-lane family
pairs = {}
for la in (0, 16, 32, 48):
    for ja in range(32):
        a = zeros.clone(); a[la, ja] = ONE
        found = []
        for qb in range(4):
            b = zeros.clone(); b[16 * qb : 16 * qb + 16, :] = ONE  # whole lane quarter of B
            if run(a, b).abs().sum() > 0:
                # which byte
                for jb in range(32):
                    b2 = zeros.clone(); b2[16 * qb : 16 * qb + 16, jb] = ONE
                    if run(a, b2).abs().sum() > 0:
                        found.append((qb, jb))
        pairs[(la, ja)] = found
bad = {k: v for k, v in pairs.items() if v != [(k[0] >> 4, k[1])]}
print("pairing (A lane quarter q, byte j) <-> (B lane quarter q, byte j) holds for all probed:", not bad)
if bad:
    for k in list(bad)[:40]:
        print("  A", k, "pairs with B (quarter, byte)", bad[k])
# scales: scale of A lane la doubled -> which outputs double
for la in (0, 16, 5, 37):
    sa = unit.clone(); sa[la] = 128
    out = run(ones, ones, sa=sa)
    ch = (out != 128).nonzero().tolist()
    print(f"scale_a of lane {la} = 2^1 -> changed outputs {sorted(set((l, e) for l, e in ch))[:12]} values {out[out != 128].unique().tolist()}")
for lb in (0, 16, 5, 37):
    sb = unit.clone(); sb[lb] = 128
    out = run(ones, ones, sb=sb)
    ch = (out != 128).nonzero().tolist()
    print(f"scale_b of lane {lb} = 2^1 -> changed outputs {sorted(set((l, e) for l, e in ch))[:12]} values {out[out != 128].unique().tolist()}")
# which K block does a lane's scale apply to: A one-hot in lane quarter qa, scale of lane quarter qs doubled
for qa in range(4):
    row = []
    for qs in range(4):
        a = zeros.clone(); a[16 * qa + 3, 7] = ONE
        sa = unit.clone(); sa[16 * qs + 3] = 128
        row.append(float(run(a, ones, sa=sa).max()))
    print(f"A one-hot in quarter {qa}: output with scale_a of quarter 0..3 doubled = {row}")
for qb in range(4):
    row = []
    for qs in range(4):
        b = zeros.clone(); b[16 * qb + 3, 7] = ONE
        sb = unit.clone(); sb[16 * qs + 3] = 128
        row.append(float(run(ones, b, sb=sb).max()))
    print(f"B one-hot in quarter {qb}: output with scale_b of quarter 0..3 doubled = {row}")
# OPSEL / scale byte: values of the other bytes of the scale dword
for byte in (1, 2, 3):
    sa = torch.full((64,), 127 | (130 << (8 * byte)), dtype=torch.int64).to(torch.int32).to(dev) if byte < 3 else torch.full((64,), 127 - (126 << 24), dtype=torch.int64).to(torch.int32).to(dev)
    print(f"scale_a dword with another value in byte {byte}: values {run(ones, ones, sa=sa).unique().tolist()}")

# ---- FP4 (e2m1) A operand against e4m3 B: which (B quarter, byte) does nibble ja of A lane quarter qa multiply
ONE4 = 0x2  # e2m1 1.0


def a4_onehot(lane, nib):
    a = zeros.clone()
    a[lane, nib // 2] = ONE4 << (4 * (nib & 1))
    return a


ones4 = zeros.clone(); ones4[:, :16] = 0x22
print("fp4 ones x fp8 ones:", run(ones4, ones, fa=4).unique().tolist())
print("fp4 ones (bytes 16..31 set too) x fp8 ones:", run(torch.full((64, 32), 0x22, dtype=torch.uint8), ones, fa=4).unique().tolist())
table = {}
for qa in range(4):
    for ja in range(32):
        a = a4_onehot(16 * qa + 2, ja)
        hit = None
        for qb in range(4):
            b = zeros.clone(); b[16 * qb : 16 * qb + 16, :] = ONE
            if run(a, b, fa=4).abs().sum() > 0:
                for jb in range(32):
                    b2 = zeros.clone(); b2[16 * qb : 16 * qb + 16, jb] = ONE
                    if run(a, b2, fa=4).abs().sum() > 0:
                        hit = (qb, jb)
        table[(qa, ja)] = hit
for qa in range(4):
    print(f"fp4 A quarter {qa}: nibble j -> B (quarter, byte):", [table[(qa, j)] for j in range(32)])
for qa in range(4):
    row = []
    for qs in range(4):
        a = a4_onehot(16 * qa + 3, 5)
        sa = unit.clone(); sa[16 * qs + 3] = 128
        row.append(float(run(a, ones, sa=sa, fa=4).max()))
    print(f"fp4 A one-hot in quarter {qa}: output with scale_a of quarter 0..3 doubled = {row}")
for qa in range(4):
    for half in (0, 1):
        row = []
        for qs in range(4):
            a = zeros.clone(); a[16 * qa + 3, 16 * half + 5] = ONE
            sa = unit.clone(); sa[16 * qs + 3] = 128
            row.append(float(run(a, ones, sa=sa).max()))
        print(f"fp8 A one-hot in quarter {qa} piece {half}: output with scale_a of quarter 0..3 doubled = {row}")
