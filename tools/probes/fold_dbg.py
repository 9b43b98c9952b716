import importlib, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import rel_err
from oracle import pna3d_oracle as O
amd = importlib.import_module('3dinfomax_amd'); po = importlib.import_module('3dinfomax_amd.pna_original'); synth = importlib.import_module('3dinfomax_amd.synth')
import test_pna_original as T
mols = synth.make_dataset(48, seed=77)
snorm = O.snorm_n([m.n_atoms for m in mols]).cuda()
res = {}
for mode in ('fold', 'blocks', 'tower'):
    po.TOWER_STACK = mode != 'tower'
    po.TOWER_FOLD = mode == 'fold'
    torch.manual_seed(5)
    model = amd.PNAOriginal(**T.PNA_ORIG_YML)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('linear.weight'):
                p.mul_(p.shape[1] * 0.5)
    model.cuda().train()
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to('cuda:0')
    out = model(g2.local_copy(), snorm)
    out.sum().backward()
    res[mode] = (out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}, g2)
for m in ('fold', 'blocks'):
    print(m, 'out rel_err', rel_err(res[m][0], res['tower'][0]))
    worst = sorted(((rel_err(v, res['tower'][1][k]), k) for k, v in res[m][1].items() if not k.endswith('posttrans.fully_connected.0.linear.bias')), reverse=True)[:6]
    print(m, 'grads worst', worst)
