#!/bin/bash
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, numpy as np
a = (torch.rand(4_000_000, dtype=torch.float32) * 6.2831855)
for name, f in (("sin", torch.sin), ("cos", torch.cos)):
    g = f(a.cuda()).cpu().numpy(); c = f(a).numpy(); n = getattr(np, name)(a.numpy().astype(np.float64)).astype(np.float32)
    print(name, "device != host(torch cpu):", float((g != c).mean()), " device != correctly rounded:", float((g != n).mean()), " host != correctly rounded:", float((c != n).mean()))
PY
