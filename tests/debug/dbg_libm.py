import numpy as np, torch
rng = np.random.default_rng(0)
x = rng.uniform(0, 2 * np.pi, 1 << 22).astype(np.float32)
y = rng.uniform(-30000, 30000, 1 << 22).astype(np.float32)
z = rng.uniform(-30000, 30000, 1 << 22).astype(np.float32)
dx, dy, dz = [torch.from_numpy(a).cuda() for a in (x, y, z)]
for name, g, c in (("sin", torch.sin(dx.double()).float(), np.sin(x.astype(np.float64)).astype(np.float32)),
                   ("cos", torch.cos(dx.double()).float(), np.cos(x.astype(np.float64)).astype(np.float32)),
                   ("atan2", torch.atan2(dy.double(), dz.double()).float(), np.arctan2(y.astype(np.float64), z.astype(np.float64)).astype(np.float32))):
    print(name, "float-rounded mismatches:", int((g.cpu().numpy() != c).sum()), "of", x.size)
gd = torch.sin(dx.double()).cpu().numpy(); cd = np.sin(x.astype(np.float64))
print("double sin: max ulp diff", np.max(np.abs(gd - cd) / np.spacing(np.abs(cd))))
gd = torch.atan2(dy.double(), dz.double()).cpu().numpy(); cd = np.arctan2(y.astype(np.float64), z.astype(np.float64))
print("double atan2: max ulp diff", np.max(np.abs(gd - cd) / np.spacing(np.abs(cd))))
