import importlib, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import rel_err
from oracle import pna3d_oracle as O
import test_pna_original as T
amd = importlib.import_module('3dinfomax_amd'); po = importlib.import_module('3dinfomax_amd.pna_original'); synth = importlib.import_module('3dinfomax_amd.synth')
ops = importlib.import_module('3dinfomax_amd.ops')
mols = synth.make_dataset(64, seed=77)
snorm = O.snorm_n([m.n_atoms for m in mols]).cuda()
res = {}
for mode in ('fp32', 'bf16'):
    ops.set_matmul_precision(mode)
    for stacked in (True, False):
        po.TOWER_STACK = stacked
        torch.manual_seed(5)
        model = amd.PNAOriginal(**T.PNA_ORIG_YML)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith('linear.weight'):
                    p.mul_(p.shape[1] * 0.5)
        model.cuda().train()
        g2 = amd.batch([amd.bond_graph(m) for m in mols]).to('cuda:0')
        out = model(g2, snorm)
        out.sum().backward()
        torch.cuda.synchronize()
        res[(mode, stacked)] = out.detach().cpu()
ops.set_matmul_precision('fp32')
print('bf16 stacked vs fp32 per-tower', rel_err(res[('bf16', True)], res[('fp32', False)]))
print('bf16 per-tower vs fp32 per-tower', rel_err(res[('bf16', False)], res[('fp32', False)]))
print('fp32 stacked vs fp32 per-tower', rel_err(res[('fp32', True)], res[('fp32', False)]))
