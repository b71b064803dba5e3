"""Builds libepos_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m epos_amd.build            # or epos_amd.build.build()
    python -m epos_amd.build --ref      # + libepos_hip_ref.so (test-only, see build_ref)

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libepos_hip.so')
# The product library plus the fp32-MFMA reference GEMM kernels of csrc/ref/ (rounds 1-2).
# Nothing of the product path needs them; the accuracy tests compare the product kernels
# against them (same C ABI, selected with EPOS_HIP_LIB or epos_amd._lib.load_ref()).
REF_LIB_PATH = os.path.join(LIB_DIR, 'libepos_hip_ref.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
         '-fPIC', '-shared']


def sources():
  return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def ref_sources():
  return sorted(glob.glob(os.path.join(CSRC, 'ref', '*.hip')))


def _stale():
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
      os.path.join(HERE, '..', 'include', '*.h'))
  return any(os.path.getmtime(d) > t for d in deps)


def build_ref(force=False, verbose=False):
  """libepos_hip_ref.so = the product library's objects + csrc/ref/*.hip (the fp32-MFMA GEMM
  kernels, which register themselves with the GEMM dispatcher when linked in). Test-only."""
  build(force=force, verbose=verbose)
  obj_dir = os.path.join(LIB_DIR, 'obj')
  headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
      os.path.join(HERE, '..', 'include', '*.h'))
  hdr_time = max(os.path.getmtime(h) for h in headers)
  cflags = [f for f in FLAGS if f != '-shared'] + ['-c', '-Wno-inline-asm']
  objs = [os.path.join(obj_dir, os.path.basename(src) + '.o') for src in sources()]
  newest = max(os.path.getmtime(o) for o in objs)
  for src in ref_sources():
    obj = os.path.join(obj_dir, 'ref_' + os.path.basename(src) + '.o')
    if (force or not os.path.exists(obj) or
        os.path.getmtime(obj) <= max(os.path.getmtime(src), hdr_time)):
      cmd = [HIPCC] + cflags + ['-o', obj, src]
      if verbose:
        print(' '.join(cmd))
      subprocess.check_call(cmd)
    objs.append(obj)
    newest = max(newest, os.path.getmtime(obj))
  if (force or not os.path.exists(REF_LIB_PATH) or os.path.getmtime(REF_LIB_PATH) <= newest):
    link = [f for f in FLAGS if f.startswith('--offload-arch') or f in ('-shared', '-fPIC')]
    cmd = [HIPCC] + link + ['-o', REF_LIB_PATH] + objs
    if verbose:
      print(' '.join(cmd))
    subprocess.check_call(cmd)
  return REF_LIB_PATH


def build_variant(name, defines, only=None):
  """A copy of the library with extra -D flags (tools/: same-box A/B runs, selected with
  EPOS_HIP_LIB=<returned path>). only = basenames of the translation units the flags matter
  for: the others are taken from the default build's object cache (build() first)."""
  out = os.path.join(LIB_DIR, 'libepos_hip_%s.so' % name)
  os.makedirs(LIB_DIR, exist_ok=True)
  if not only:
    subprocess.check_call([HIPCC] + FLAGS + ['-Wno-inline-asm'] + list(defines) + ['-o', out] +
                          sources())
    return out
  build()
  obj_dir = os.path.join(LIB_DIR, 'obj')
  cflags = [f for f in FLAGS if f != '-shared'] + ['-c', '-Wno-inline-asm']
  objs = []
  for src in sources():
    base = os.path.basename(src)
    if base in only:
      obj = os.path.join(obj_dir, '%s.%s.o' % (base, name))
      subprocess.check_call([HIPCC] + cflags + list(defines) + ['-o', obj, src])
    else:
      obj = os.path.join(obj_dir, base + '.o')
    objs.append(obj)
  link = [f for f in FLAGS if f.startswith('--offload-arch') or f in ('-shared', '-fPIC')]
  subprocess.check_call([HIPCC] + link + ['-o', out] + objs)
  return out


def build(force=False, verbose=False):
  """Compiles every HIP source into epos_amd/lib/libepos_hip.so: one hipcc job per
  translation unit (in parallel; objects cached under lib/obj/ by mtime AND by the compile
  flags they were built with), then a link with the same target flags."""
  if not force and not _stale() and _flags_stamp_ok():
    return LIB_PATH
  from concurrent.futures import ThreadPoolExecutor
  obj_dir = os.path.join(LIB_DIR, 'obj')
  os.makedirs(obj_dir, exist_ok=True)
  headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
      os.path.join(HERE, '..', 'include', '*.h'))
  hdr_time = max(os.path.getmtime(h) for h in headers)
  cflags = [f for f in FLAGS if f != '-shared'] + ['-c', '-Wno-inline-asm']
  flags_changed = not _flags_stamp_ok()

  def compile_one(src):
    obj = os.path.join(obj_dir, os.path.basename(src) + '.o')
    if (not force and not flags_changed and os.path.exists(obj) and
        os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time)):
      return obj
    cmd = [HIPCC] + cflags + ['-o', obj, src]
    if verbose:
      print(' '.join(cmd))
    subprocess.check_call(cmd)
    return obj
  with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
    objs = list(ex.map(compile_one, sources()))
  link = [f for f in FLAGS if f.startswith('--offload-arch') or f in ('-shared', '-fPIC')]
  cmd = [HIPCC] + link + ['-o', LIB_PATH] + objs
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd)
  with open(_STAMP, 'w') as f:
    f.write(_flags_key())
  return LIB_PATH


_STAMP = os.path.join(LIB_DIR, 'obj', 'flags.stamp')


def _flags_key():
  return ' '.join([HIPCC] + FLAGS)


def _flags_stamp_ok():
  try:
    with open(_STAMP) as f:
      return f.read() == _flags_key()
  except OSError:
    return False


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True))
  if '--ref' in sys.argv:
    print(build_ref(verbose=True))
