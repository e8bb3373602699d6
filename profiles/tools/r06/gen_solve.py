#!/usr/bin/env python3
"""Generates the written-out lone-lane sweep loops of quad_floor_solve_n (quadx_fast.hpp) -- C++ source text with inline asm.
N = 1: unrolled by two sweeps (the impulses alternate between two register sets: no copy), 20 instructions a sweep.
N = 2: the six rows' velocities as three aligned pairs in fixed registers, every row's five coupling updates as three v_pk_fma_f32."""
import sys

def n1():
    def sweep(o, n, exit_lbl):
        L = []
        a = L.append
        a(f"v_max_f32 %[{n}0], 0, %[e0]")
        a(f"v_sub_f32 %[d0], %[{n}0], %[{o}0]")
        a("v_fma_f32 %[e1], -%[b10], %[d0], %[e1]")
        a("v_fma_f32 %[e2], -%[b20], %[d0], %[e2]")
        a(f"v_mul_f32 %[lim], %[fx], %[{n}0]")
        a(f"v_med3_f32 %[{n}1], %[e1], -%[lim], %[lim]")
        a(f"v_sub_f32 %[d1], %[{n}1], %[{o}1]")
        a("v_fma_f32 %[e2], -%[b21], %[d1], %[e2]")
        a("v_fma_f32 %[e0], -%[b01], %[d1], %[e0]")
        a(f"v_mul_f32 %[lim], %[fy], %[{n}0]")
        a(f"v_med3_f32 %[{n}2], %[e2], -%[lim], %[lim]")
        a(f"v_sub_f32 %[d2], %[{n}2], %[{o}2]")
        a("v_fma_f32 %[e0], -%[b02], %[d2], %[e0]")
        a("v_fma_f32 %[e1], -%[b12], %[d2], %[e1]")
        a("v_max3_f32 %[lim], |%[d0]|, |%[d1]|, |%[d2]|")
        a("v_cmp_lt_f32 vcc, %[bound], %[lim]")
        a("s_and_b64 vcc, vcc, %[on]")
        a("s_sub_u32 %[cnt], %[cnt], 1")
        return L
    L = ["1:"] + sweep("l", "n", None) + ["s_cbranch_vccz 3f", "s_cbranch_scc1 3f"] + sweep("n", "l", None) + ["s_cbranch_vccz 2f", "s_cbranch_scc0 1b", "s_branch 2f",
         "3:", "v_mov_b32 %[l0], %[n0]", "v_mov_b32 %[l1], %[n1]", "v_mov_b32 %[l2], %[n2]", "2:"]
    return L

E0 = 232  # e0 .. e5 = v232 .. v237 (three aligned pairs); (d, 1.0) pairs: v[238:239], v[240:241]
def ereg(i): return f"v{E0 + i}"
def epair(k): return f"v[{E0 + 2 * k}:{E0 + 2 * k + 1}]"
DP = [("v238", "v[238:239]"), ("v240", "v[240:241]")]

def n2():
    def sweep(o, n):
        L = []
        a = L.append
        for r in range(6):
            c, d = divmod(r, 3)
            dreg, dpair = DP[r & 1]
            if d == 0:
                a(f"v_max_f32 %[{n}{r}], 0, {ereg(r)}")
            else:
                a(f"v_mul_f32 %[lim], %[f{'xy'[d - 1]}{c}], %[{n}{3 * c}]")
                a(f"v_med3_f32 %[{n}{r}], {ereg(r)}, -%[lim], %[lim]")
            a(f"v_sub_f32 {dreg}, %[{n}{r}], %[{o}{r}]")
            k = r // 2
            # the pair that holds the row itself first? no: the NEXT row's pair first (the dependent chain waits for it)
            order = [((r + 1) % 6) // 2] + [x for x in range(3) if x != ((r + 1) % 6) // 2]
            for kk in order:
                if kk == k:  # half update: the row itself passes through exactly ((-0) * 1.0 + e = e)
                    sel = "op_sel:[0,1,0] op_sel_hi:[1,0,1] " if r % 2 == 0 else ""
                    a(f"v_pk_fma_f32 {epair(kk)}, %[bh{r}], {dpair}, {epair(kk)} {sel}neg_lo:[1,0,0] neg_hi:[1,0,0]")
                else:
                    a(f"v_pk_fma_f32 {epair(kk)}, %[bw{r}{kk}], {dpair}, {epair(kk)} op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
            if r == 1:
                a(f"v_max_f32 %[t], |{DP[0][0]}|, |{DP[1][0]}|")
            elif r & 1:
                a(f"v_max3_f32 %[t], %[t], |{DP[0][0]}|, |{DP[1][0]}|")
        a("v_cmp_lt_f32 vcc, %[bound], %[t]")
        a("s_and_b64 vcc, vcc, %[on]")
        a("s_sub_u32 %[cnt], %[cnt], 1")
        return L
    L = ["v_mov_b32 v239, 1.0", "v_mov_b32 v241, 1.0", "1:"] + sweep("l", "n") + ["s_cbranch_vccz 3f", "s_cbranch_scc1 3f"] + sweep("n", "l") + \
        ["s_cbranch_vccz 2f", "s_cbranch_scc0 1b", "s_branch 2f", "3:"] + [f"v_mov_b32 %[l{r}], %[n{r}]" for r in range(6)] + ["2:"]
    return L

def cstr(L, ind):
    out = []
    for i, x in enumerate(L):
        last = i == len(L) - 1
        if x.endswith(":"):
            out.append(f'{ind}"{x}' + ('"' if last else '\\n\\t"'))
        else:
            out.append(f'{ind}"{x}' + ('"' if last else '\\n\\t"'))
    return "\n".join(out)

if __name__ == "__main__":
    which = sys.argv[1]
    L = n1() if which == "1" else n2()
    body = [x for x in L if not x.endswith(":")]
    print(f"// {len(L)} lines; per sweep: {(len([x for x in (n1() if which=='1' else n2()) if not x.endswith(':')]) )}", file=sys.stderr)
    print(cstr(L, "        "))
