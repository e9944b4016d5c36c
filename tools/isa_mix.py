#!/usr/bin/env python3
"""Instruction mix of the fused kernel's frame loop, from the compiler's own assembly.

Compiles rtl-power-fftw_amd/csrc/rpf_kernels.hip to gfx950 assembly (device only, the
flags of the shipped build), finds the default (no window, LDS-DMA) instantiation of
fft_accum_kernel for each N, takes its hottest loop (the backward branch spanning the
most instructions = the per-frame loop) and prices every instruction with the
cycles-per-wave-instruction tables of MI355X_MICROARCH.md:

  VALU on a SIMD-32: 2 cycles per wave64 instruction; packed-f32 (v_pk_*_f32), every f64
  instruction and v_cvt_f64_f32 / v_cvt_f32_f64 issue at half rate: 4 cycles.
  LDS: ds_read_b32/u16/b64 2, ds_read_b128 4, ds_write_b32 4, ds_write_b64 6,
  ds_write_b128 13 (cycles of the CU's LDS path per wave-instruction).

Writes profiles/isa_mix.json, which bench.py turns into `roofline.secondary`
(VALU issue-slot fraction, LDS fraction) for the kernel time it measures.
Runs in the build container (no GPU needed):  python tools/isa_mix.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtl-power-fftw_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]

LDS_CYCLES = {"ds_read_b32": 2, "ds_read_u16": 2, "ds_read_u8": 2, "ds_read_b64": 2, "ds_read_b128": 4,
              "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_read_u16_d16": 2, "ds_read_u16_d16_hi": 2,
              "ds_write_b32": 4, "ds_write_b64": 6, "ds_write_b128": 13, "ds_write2_b32": 6,
              "ds_write2_b64": 13, "ds_write_b16": 4}


def valu_cycles(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return 4
    if op.endswith("_f64") or op in ("v_cvt_f64_f32", "v_cvt_f32_f64"):
        return 4
    return 2


def kernel_bodies(asm):
    """name -> list of instruction lines for every fft_accum_kernel instantiation."""
    out = {}
    cur = None
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN3rpf\S*fft_accum_kernel\S*):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(line)
    return out


def hottest_loop(lines):
    """Instructions of the per-frame loop (label ... backward branch)."""
    labels, instrs = {}, []
    for line in lines:
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        instrs.append(t.split(";")[0].strip())
    # the frame loop = the backward branch whose span holds the most packed-f32 arithmetic; among
    # nested loops with the same arithmetic (the per-hop segment loop encloses the frame loop and adds
    # only the hand-over) the shortest span
    best = None
    for idx, ins in enumerate(instrs):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= idx:
            lo = labels[m.group(1)]
            pk = sum(1 for i in instrs[lo: idx + 1] if i.startswith("v_pk_"))
            key = (pk, -(idx - lo))
            if best is None or key > best[0]:
                best = (key, lo, idx)
    return instrs[best[1]: best[2] + 1] if best else []


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out_s = os.path.join(tmp, "k1.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + FLAGS +
                       ["-S", "--cuda-device-only", "-o", out_s, os.path.join(CSRC, "rpf_kernels.hip")],
                       check=True)
        asm = open(out_s).read()
    result = {}
    for name, lines in kernel_bodies(asm).items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        m = re.search(r"fft_accum_kernel<rpf::Geom<(\d+), (\d+)>, (\d+), (\d+), (false|true), (false|true)", dem)
        if not m:
            continue
        N, P, WG, occ, windowed, dma = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), \
            m.group(5) == "true", m.group(6) == "true"
        if windowed or not dma:
            continue
        loop = hottest_loop(lines)
        counts, valu, lds = {}, 0, 0
        for ins in loop:
            op = ins.split()[0]
            counts[op] = counts.get(op, 0) + 1
            if op.startswith("v_"):
                valu += valu_cycles(op)
            elif op.startswith("ds_"):
                lds += LDS_CYCLES.get(op, 4)
        T = N // P
        result[str(N)] = {
            "kernel": "fft_accum_kernel<Geom<%d,%d>,%d,%d,window=false,dma=true>" % (N, P, WG, occ),
            "waves": WG // 64, "frames_per_iteration": WG // T,
            "loop_instructions": len(loop),
            "valu_instructions": sum(v for k, v in counts.items() if k.startswith("v_")),
            "valu_cycles_per_wave_iteration": valu,
            "lds_instructions": sum(v for k, v in counts.items() if k.startswith("ds_")),
            "lds_cycles_per_wave_iteration": lds,
            "top_ops": dict(sorted(counts.items(), key=lambda kv: -kv[1])[:14]),
        }
        print(result[str(N)]["kernel"], "loop", len(loop), "instr; VALU", valu, "cycles, LDS", lds, "cycles per wave-iteration")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(result, open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
