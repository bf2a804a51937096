"""Cycle shares of the single-workgroup Lanczos kernel (diagnostic build):
  VTC_LANCZOS_STAMPS=1 python3 tools/lanczos_stamps.py"""
import os, sys
os.environ['VTC_LANCZOS_STAMPS'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vision-transform-codes_amd'))
import numpy as np, torch
import vtc_hip
dev = torch.device('cuda:0')
lib = vtc_hip.load_library()
rs = np.random.RandomState(0)
for s, n in ((256, 256), (1024, 256), (128, 121)):
  D = rs.randn(s, n).astype(np.float32)
  D /= np.sqrt((D ** 2).sum(1))[:, None]
  G = vtc_hip.gram(torch.from_numpy(D).to(dev), transpose_a=True)
  out = torch.zeros(16, dtype=torch.float32, device=dev)
  for _ in range(3):
    vtc_hip.check(lib.vtc_lambda_max(vtc_hip.ptr(G), n, vtc_hip.ptr(out), None, 0,
                                     vtc_hip.current_stream(dev)), 'lanczos')
  o = out.cpu().numpy()
  names = ['A matvec', 'B update', 'bounds', 'solve']
  print('n=%d s=%d lambda %.6f (eigvalsh %.6f) total %.0f cycles' % (
      n, s, o[0], float(torch.linalg.eigvalsh(G.double())[-1]), o[3:7].sum()))
  for k, name in enumerate(names):
    print('   %-10s %9.0f cycles' % (name, o[3 + k]))
