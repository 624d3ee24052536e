"""Diagnostic: device vs f64-oracle gradient error of the config-3 step, by parameter group, spatial_sort on/off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_step_parity as T
dev = torch.device("cuda:0")
for ss in (False, 5):
    for fused in ("1", "0"):
        import unscene3d_amd.models.criterion as CR
        CR.FUSED = fused == "1"
        cfg, batch, collate, module = T._setup(dev, ss)
        data, target, names = collate(batch)
        module.model.randperm = T.PermSource()
        total, _ = module.training_step((data, target, names))
        total.backward()
        res = {}
        for dt in (torch.float32, torch.float64):
            sd = T._leaves(module, dt)
            tot, _ = T._oracle_step(module, cfg, sd, data, target, T.PermSource(), dt)
            tot.backward()
            res[dt] = sd
        groups = {}
        rows = []
        for name, p in module.model.named_parameters():
            if name.startswith("backbone.final."): continue
            g64 = res[torch.float64][name].grad
            if float(g64.norm()) < 1e-12: continue
            grp = ".".join(name.split(".")[:2])
            a = groups.setdefault(grp, [0.0, 0.0, 0.0])
            d = float((p.grad.double().cpu() - g64).square().sum()); c = float((res[torch.float32][name].grad.double() - g64).square().sum()); n = float(g64.square().sum())
            a[0] += d; a[1] += c; a[2] += n
            rows.append((T.rel_err(p.grad, g64), T.rel_err(res[torch.float32][name].grad, g64), name))
        D = sum(a[0] for a in groups.values()); C = sum(a[1] for a in groups.values()); N = sum(a[2] for a in groups.values())
        print(f"== spatial_sort={ss} fused={fused}: loss {float(total):.6f}  global dev {(D/N)**0.5:.2e} cpu32 {(C/N)**0.5:.2e}")
        for g, a in groups.items():
            print(f"   {g:32s} dev {(a[0]/a[2])**0.5:.2e} cpu {(a[1]/a[2])**0.5:.2e}  |g| {a[2]**0.5:.3e}")
        rows.sort(reverse=True)
        for r in rows[:6]: print("   worst", f"{r[0]:.2e} (cpu {r[1]:.2e})", r[2])
