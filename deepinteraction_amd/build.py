"""Build libdeepinteraction_hip.so (gfx950) in-tree with hipcc.

`python -m deepinteraction_amd.build` or `__graft_entry__.build()`.  hipcc
cross-compiles without a GPU; the resulting .so is git-ignored but travels with
the tree to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdeepinteraction_hip.so')
ARCH = 'gfx950'
# No packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): on the MI355X boxes of this project
# they returned wrong results in lanes 48-63 - sporadically, only while a matrix-core kernel shared the CU (two streams /
# two samples in flight) - see DESIGN.md section 5, tools/hazard/ and tools/debug/keys_race3.py.  Scalar fp32 ops cost the hot kernels nothing
# measurable (they are bound by MFMA, LDS or memory).
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-cuda-compat', '-Wno-unused-result', '-Wno-inline-asm',
         f'--offload-arch={ARCH}', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


# per-file flags; DI_PACKED_FP32=1 (experiments only) compiles WITH the packed instructions
# i2p_dense.hip: MFMA results in arch VGPRs (the accumulators are touched by a rare VALU rescale; with AGPR accumulators
# the compiler moves all 32 of them to VGPRs and back around every block - 64 moves per 12 MFMAs)
EXTRA = {'i2p_dense.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
if os.environ.get('DI_PACKED_FP32') == '1':
    FLAGS = [f for f in FLAGS if f not in ('-Xclang', '-target-feature', '-packed-fp32-ops')]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def up_to_date():
    if not os.path.exists(LIB):
        return False
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in sources():      # one hipcc per translation unit, in parallel
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = [hipcc] + [f for f in FLAGS if f != '-shared'] + EXTRA.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + out.decode())
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
