"""Probe: is torch's CPU float32 ``log`` the same function on this host as on the fixture host?
Prints a digest of torch.log over a fixed 2^24-sample of (2^-40, 2) plus the mismatch count against the correctly
rounded value (float64 log rounded once).  Run here and on the GPU box; compare the lines."""
import hashlib
import numpy as np
import torch

torch.manual_seed(0)
lo, hi = int(np.float32(2.0 ** -40).view(np.int32)), int(np.float32(2.0).view(np.int32))
x = torch.randint(lo, hi, (1 << 24,), dtype=torch.int32).view(torch.float32)
a = torch.log(x)
cr = torch.log(x.double()).float()
d = a.view(torch.int32) - cr.view(torch.int32)
print('torch', torch.__version__, 'threads', torch.get_num_threads())
print('log digest', hashlib.sha256(a.numpy().tobytes()).hexdigest()[:16], 'vs CR: -1 ulp %d, +1 ulp %d, other %d of %d'
      % (int((d == -1).sum()), int((d == 1).sum()), int((d.abs() > 1).sum()), d.numel()))
s = torch.sigmoid(torch.randn(1 << 22, generator=torch.Generator().manual_seed(1)) * 3 - 4)
print('sigmoid digest', hashlib.sha256(s.numpy().tobytes()).hexdigest()[:16])
for name, y in (('log(p+eps)', s + 1e-12), ('log(1-p+eps)', 1 - s + 1e-12)):
    a = torch.log(y); cr = torch.log(y.double()).float()
    print(name, 'digest', hashlib.sha256(a.numpy().tobytes()).hexdigest()[:16], 'mismatch vs CR %d of %d' % (int((a != cr).sum()), a.numel()))
import subprocess
print(subprocess.run("lscpu | grep -iE 'model name|^CPU\\(s\\)|Thread|Core|Socket'", shell=True, capture_output=True, text=True).stdout)
