"""Scans the gfx950 ISA of every translation unit of csrc/ for the store pattern
hipcc (ROCm 7.2) mis-handles: a 12- or 16-byte MUBUF store whose scalar offset
is a REGISTER.  For that form the compiler inserts no wait state between the
store and a following VALU write of its data registers (it believes the hazard
exists only with an immediate offset); on MI355X the store then sometimes
carries the new value.  The sources keep the scalar offset of
wide buffer stores at the constant 0; this is the check on what the compiler
actually emitted.  No GPU needed (cross-compiles).

  python3 tools/check_store_hazard.py            # exit code 1 if any is found
"""
import pathlib
import re
import subprocess
import sys
import tempfile

REPO = pathlib.Path(__file__).resolve().parent.parent
CSRC = REPO / 'vision-transform-codes_amd' / 'csrc'
# buffer_store_dwordx4 v[26:29], v30, s[8:11], s77 offen ...
WIDE = re.compile(r'^\s*buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*([^,]+),\s*'
                  r's\[\d+:\d+\],\s*(\S+)')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
         '--cuda-device-only', '-S', '-I', str(REPO / 'include')]
NO_SLP = {'fc_fused.hip', 'conv.hip'}


def main():
  bad_total = 0
  with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(CSRC.glob('*.hip')):
      out = pathlib.Path(tmp) / (src.stem + '.s')
      flags = FLAGS + (['-fno-slp-vectorize'] if src.name in NO_SLP else [])
      subprocess.run(['/opt/rocm/bin/hipcc'] + flags + [str(src), '-o',
                                                        str(out)],
                     check=True, stderr=subprocess.DEVNULL)
      wide = bad = 0
      kernel = ''
      where = []
      for line in out.read_text().splitlines():
        if line.endswith(':') and line.startswith('_Z'):
          kernel = line[:-1]
        m = WIDE.match(line)
        if not m:
          continue
        wide += 1
        soffset = m.group(2)
        if re.fullmatch(r's\d+|m0|s\[\d+\]', soffset):
          bad += 1
          where.append((kernel[:60], line.strip()))
      print('%-22s %5d wide buffer stores, %d with a register scalar offset'
            % (src.name, wide, bad))
      for k, l in where[:5]:
        print('    %s: %s' % (k, l))
      bad_total += bad
  print('TOTAL with a register scalar offset: %d' % bad_total)
  return 1 if bad_total else 0


if __name__ == '__main__':
  sys.exit(main())
