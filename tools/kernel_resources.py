"""Static resource table of every kernel in the built library (no GPU needed): architectural + accumulation VGPRs, SGPRs,
scratch (spill) bytes, static LDS, the workgroup size bound, and the residency those imply on gfx950 (512 VGPRs per SIMD
lane shared by the resident waves in steps of 8, at most 8 waves per SIMD, 160 KB of LDS per CU).

Reads the `.hip_fatbin` section of `deepinteraction_amd/libdeepinteraction_hip.so` (uncompressed clang offload bundles, one
per translation unit), takes the gfx950 code objects and lets `llvm-readelf --notes` decode their AMDGPU metadata.

    python tools/kernel_resources.py [--markdown profiles/rNN_kernel_resources.md] [--min-vgpr 0]
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([f'{LLVM}/llvm-objcopy', f'--dump-section=.hip_fatbin={fat}', lib, os.path.join(tmp, 'copy.so')], check=True)
        data = open(fat, 'rb').read()
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from('<QQQ', data, pos)
            ident = data[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if 'gfx950' in ident and size:
                yield data[base + off:base + off + size]


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(elf_bytes)
        f.flush()
        text = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', f.name], check=True, capture_output=True, text=True).stdout
    # the metadata prints as YAML: one "- .agpr_count: ..." item per kernel under amdhsa.kernels
    for block in re.split(r'\n\s*- \.', text)[1:]:
        fields = dict(re.findall(r'^\s*\.?([a-z_]+):\s+(.+?)\s*$', '.' + block, flags=re.M))
        if 'vgpr_count' not in fields or 'name' not in fields:
            continue
        yield fields


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r'^void ', '', re.sub(r'\(.*$', '', o)) for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'deepinteraction_amd', 'libdeepinteraction_hip.so'))
    ap.add_argument('--markdown')
    ap.add_argument('--min-vgpr', type=int, default=0)
    args = ap.parse_args()
    rows = []
    for co in code_objects(args.lib):
        for k in kernels_of(co):
            arch, acc = int(k['vgpr_count']), int(k.get('agpr_count', 0))
            # `.vgpr_count` is the unified count on gfx90a+ (architectural + accumulation, the allocation unit is 8)
            total = max(arch, 1)
            waves_simd = min(8, 512 // ((total + 7) // 8 * 8))
            wg = int(k.get('max_flat_workgroup_size', 1024))
            waves_wg = (wg + 63) // 64
            lds = int(k.get('group_segment_fixed_size', 0))
            by_regs = waves_simd * 4 // waves_wg if waves_wg <= waves_simd * 4 else 0
            by_lds = (160 * 1024) // lds if lds else 99
            rows.append(dict(name=k['name'].strip("'\""), vgpr=arch, agpr=acc, sgpr=int(k['sgpr_count']),
                             scratch=int(k.get('private_segment_fixed_size', 0)), spill=int(k.get('vgpr_spill_count', 0)),
                             lds=lds, wg=wg, waves_simd=waves_simd, wg_cu=min(by_regs, by_lds, 32)))
    names = demangle([r['name'] for r in rows])
    for r, n in zip(rows, names):
        r['name'] = n
    rows = [r for r in rows if r['vgpr'] >= args.min_vgpr]
    rows.sort(key=lambda r: (-r['vgpr'], r['name']))
    head = '| kernel | VGPR (of which AGPR) | SGPR | scratch B | spilled VGPRs | static LDS B | max WG | waves / SIMD by registers | WG / CU (registers, static LDS) |'
    lines = [head, '|---|---|---|---|---|---|---|---|---|']
    for r in rows:
        lines.append(f"| `{r['name']}` | {r['vgpr']} ({r['agpr']}) | {r['sgpr']} | {r['scratch']} | {r['spill']} | {r['lds']} | "
                     f"{r['wg']} | {r['waves_simd']} | {r['wg_cu']} |")
    spilled = [r['name'] for r in rows if r['scratch'] or r['spill']]
    summary = (f"{len(rows)} kernels; {len(spilled)} with scratch / spills" + (': ' + ', '.join(f'`{s}`' for s in spilled) if spilled else ''))
    text = '\n'.join(lines) + '\n\n' + summary + '\n'
    if args.markdown:
        with open(args.markdown, 'w') as f:
            f.write('# Static kernel resources of `libdeepinteraction_hip.so` (gfx950)\n\n'
                    'From the code objects\' AMDGPU metadata (`tools/kernel_resources.py`; dynamic LDS passed at launch is not in '
                    '"static LDS", so WG / CU is an upper bound for kernels that use it).\n\n' + text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main()
