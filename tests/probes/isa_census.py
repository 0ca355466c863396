#!/usr/bin/env python3
"""Static instruction census of the render kernels from `hipcc --cuda-device-only -S` output.

usage: isa_census.py file.s [substring-of-mangled-name ...]
Prints, per kernel: VALU / SALU / SMEM / VMEM / LDS instruction counts, the share of VALU that only moves data
(v_mov*, v_cndmask*, v_readlane/v_writelane/v_readfirstlane, v_accvgpr*), the IEEE-division expansions (v_div_fixup)
and the resource metadata (VGPRs, SGPRs, spills, scratch, LDS, kernarg size)."""
import re, sys, json, collections

def parse(path):
    kernels = {}
    cur = None
    meta = {}
    with open(path) as f:
        lines = f.readlines()
    i = 0
    name_re = re.compile(r'^(_Z\w+):')
    for ln in lines:
        m = name_re.match(ln)
        if m and cur is None:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            if ln.startswith('.Lfunc_end'):
                cur = None
                continue
            s = ln.strip()
            if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
                continue
            kernels[cur].append(s.split()[0])
    # metadata (amdhsa.kernels yaml)
    txt = ''.join(lines)
    for blk in re.split(r'\n  - ', txt[txt.find('amdhsa.kernels:'):] if 'amdhsa.kernels:' in txt else ''):
        nm = re.search(r'\.name:\s+(\S+)', blk)
        if not nm:
            continue
        d = {}
        for key in ('sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'vgpr_spill_count', 'agpr_count', 'private_segment_fixed_size',
                    'group_segment_fixed_size', 'kernarg_segment_size', 'max_flat_workgroup_size'):
            mm = re.search(r'\.%s:\s+(\d+)' % key, blk)
            if mm:
                d[key] = int(mm.group(1))
        meta[nm.group(1)] = d
    return kernels, meta

MOVES = ('v_mov_b32', 'v_mov_b64', 'v_cndmask_b32', 'v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32',
         'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 'v_accvgpr_mov_b32', 'v_swap_b32')

def census(ops):
    c = collections.Counter(ops)
    valu = sum(n for k, n in c.items() if k.startswith('v_'))
    salu = sum(n for k, n in c.items() if k.startswith('s_') and not k.startswith(('s_load', 's_buffer_load', 's_waitcnt', 's_nop', 's_endpgm', 's_barrier', 's_sleep')))
    smem = sum(n for k, n in c.items() if k.startswith(('s_load', 's_buffer_load')))
    vmem = sum(n for k, n in c.items() if k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')))
    lds = sum(n for k, n in c.items() if k.startswith('ds_'))
    def fam(prefix):
        return sum(n for k, n in c.items() if k == prefix or k.startswith(prefix + '_e') or k.startswith(prefix + '_dpp') or k.startswith(prefix + '_sdwa'))
    moves = sum(fam(k) for k in MOVES)
    spill_lane = fam('v_readlane_b32') + fam('v_writelane_b32')
    out = dict(total=len(ops), valu=valu, salu=salu, smem=smem, vmem=vmem, lds=lds, waitcnt=c['s_waitcnt'], moves=moves,
               v_mov=fam('v_mov_b32') + fam('v_mov_b64'), v_cndmask=fam('v_cndmask_b32'), lane_spill=spill_lane, scratch=sum(n for k, n in c.items() if k.startswith('scratch_')),
               div=fam('v_div_fixup_f32'), div64=fam('v_div_fixup_f64'), rcp=fam('v_rcp_f32'), sqrt=fam('v_sqrt_f32'), branches=sum(n for k, n in c.items() if k.startswith('s_cbranch')),
               saveexec=sum(n for k, n in c.items() if 'saveexec' in k))
    out['useful_valu_share'] = round(1.0 - moves / valu, 4) if valu else None
    # issue-cost classes (profiles/r03_valu_issue.txt, tests/probes/microbench/valu_issue.hip, 4 waves per SIMD): a plain 32-bit
    # VALU instruction occupies its SIMD for 2.5 cycles (profiles/r04_valu_issue.txt: in-kernel cycle counters, four waves per SIMD), 32-bit integer multiplies / v_mad_u64_u32 / packed-f32 for 4.25,
    # v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos for 8.25
    half = sum(n for k, n in c.items() if re.match(r'v_(mul_lo_u32|mul_lo_i32|mul_hi_u32|mul_hi_i32|mul_u32_u24|mul_i32_i24|mad_u32_u24|mad_i32_i24|mad_u64_u32|mad_i64_i32|pk_)', k))
    trans = sum(n for k, n in c.items() if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', k))
    out['valu_half_rate'] = half
    out['valu_transcendental'] = trans
    out['valu_issue_cycles_static_mix'] = round(((valu - half - trans) * 2.5 + half * 4.25 + trans * 8.25) / valu, 4) if valu else None
    return out

def write_committed_census():
    """profiles/valu_census.json: the census of every render kernel of the CURRENT sources (compiles rt_render.hip to
    assembly with the Makefile's flags, ~80 s), keyed by the kernel name rt_last_launch_info / rocprofv3 print and tied to
    the hash of the kernel sources.  bench.py reads roofline.useful_valu_share from it."""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    import bench
    csrc = os.path.join(root, "raytracing-rust_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rt_render.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                        "-fno-gpu-rdc", "-Wno-unused-function", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause", "--cuda-device-only", "-S",
                        os.path.join(csrc, "rt_render.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        ks, meta = parse(out)
    names = [k for k in ks if "render_kernel" in k]
    pretty = subprocess.run(["c++filt"] + names, check=True, capture_output=True, text=True).stdout.splitlines()
    kernels = {}
    for mangled, p in zip(names, pretty):
        p = re.sub(r"\(.*$", "", p)
        p = p[5:] if p.startswith("void ") else p
        d = census(ks[mangled])
        d.update(meta.get(mangled, {}))
        kernels[p] = d
    path = os.path.join(root, "profiles", "valu_census.json")
    json.dump({"source_hash": bench.source_hash(), "how": "python tests/probes/isa_census.py --write",
               "note": "static instruction census of the build's gfx950 assembly, cold paths included; useful_valu_share = 1 - "
                       "(v_mov + v_cndmask + v_readlane/v_writelane/v_readfirstlane) / VALU", "kernels": kernels}, open(path, "w"), indent=1, sort_keys=True)
    print(f"wrote {path}: {len(kernels)} kernels")


if __name__ == '__main__':
    if sys.argv[1:] == ['--write']:
        write_committed_census()
        sys.exit(0)
    ks, meta = parse(sys.argv[1])
    pats = sys.argv[2:]
    res = {}
    for k, ops in ks.items():
        if pats and not any(p in k for p in pats):
            continue
        d = census(ops)
        d.update(meta.get(k, {}))
        res[k] = d
    print(json.dumps(res, indent=1))
