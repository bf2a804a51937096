"""Board power and shader clock while the fused FISTA kernel runs (read-only
rocm-smi queries from a side thread), at full and at partial occupancy of the
chip.  Backs the statement of DESIGN.md 4.1 that the headline launch runs at
the power cap.

  python3 tools/power_probe.py [seconds per case]
"""
import pathlib
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / 'vision-transform-codes_amd'))


def poll(stop, samples):
  while not stop.is_set():
    try:
      out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'],
                           capture_output=True, text=True, timeout=10).stdout
    except Exception as e:   # no rocm-smi, no permission: report and go on
      samples.append(('error', str(e)))
      return
    power = re.search(r'[Pp]ower[^\n]*?:\s*([0-9.]+)', out)
    sclk = re.search(r'sclk clock level[^\n]*?\((\d+)Mhz\)', out)
    samples.append((float(power.group(1)) if power else None,
                    int(sclk.group(1)) if sclk else None))
    time.sleep(0.3)


def main():
  from analysis_transforms.fully_connected import ista_fista
  seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
  dev = torch.device('cuda:0')
  rs = np.random.RandomState(0)
  D = rs.randn(1024, 256).astype(np.float32)
  D /= np.linalg.norm(D, axis=1, keepdims=True)
  D = torch.from_numpy(D).to(dev)
  print('%-34s %10s %12s %12s %14s' % ('case', 'ms/launch', 'ms/wg/200it',
                                       'power W', 'sclk MHz'))
  for name, batch, precision in (('all 256 CUs, f16x3', 131072, 'f16x3'),
                                 ('all 256 CUs, bf16x3', 131072, 'bf16x3'),
                                 ('all 256 CUs, bf16', 131072, 'bf16'),
                                 ('128 CUs (batch 4096), f16x3', 4096, 'f16x3'),
                                 ('32 CUs (batch 1024), f16x3', 1024, 'f16x3')):
    X = torch.from_numpy((0.1 * rs.randn(batch, 256)).astype(np.float32)).to(dev)
    iters = 200 if batch > 8192 else 2000
    run = lambda: ista_fista.run(X, D, 0.008, iters, variant='fista',
                                 precision=precision, stepsize=0.05)
    run()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    side = threading.Thread(target=poll, args=(stop, samples))
    side.start()
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < seconds:
      run()
      torch.cuda.synchronize()
      launches += 1
    elapsed = time.perf_counter() - t0
    stop.set()
    side.join()
    good = [s for s in samples[1:] if s[0] != 'error']
    power = [s[0] for s in good if s[0] is not None]
    clock = [s[1] for s in good if s[1] is not None]
    ms = 1e3 * elapsed / launches
    rounds = max(1, batch // 32 // 256)
    print('%-34s %10.1f %12.2f %12s %14s' % (
        name, ms, ms / rounds * 200.0 / iters,
        '%.0f' % np.mean(power) if power else 'n/a',
        '%.0f' % np.mean(clock) if clock else 'n/a'))
    if samples and samples[0][0] == 'error':
      print('  (rocm-smi: %s)' % samples[0][1])


if __name__ == '__main__':
  main()
