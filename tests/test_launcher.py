"""The zero-edit launcher (launch_reference.py): a script with the reference train.py's structure - star-imports, plugin
lookups through `globals()[name]` at call time (reference train.py:167-172, 189, 208-209, 589-590), everything started
from an `if __name__ == '__main__':` block - must end up with the MI355X classes, the file untouched."""
import importlib
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_rebinds_plugin_names_without_editing_the_script(tmp_path):
    pkg = tmp_path / 'models'
    pkg.mkdir()
    (pkg / '__init__.py').write_text('from models.pna import PNA\nfrom models.net3d import Net3D\n')
    (pkg / 'pna.py').write_text('class PNA:\n    origin = "reference"\n')
    (pkg / 'net3d.py').write_text('class Net3D:\n    origin = "reference"\n')
    (tmp_path / 'commons_losses.py').write_text('class NTXent:\n    origin = "reference"\nclass Other:\n    origin = "reference"\n')
    (tmp_path / 'collates.py').write_text('def contrastive_collate(b):\n    return "reference"\n')
    script = tmp_path / 'train.py'
    script.write_text(textwrap.dedent('''
        import json, sys
        from collates import *
        from models import *
        from torch.optim import *
        from commons_losses import *

        def get_trainer(args):
            return globals()[args['model3d_type']], globals()[args['loss_func']], globals()[args['optimizer']]

        def load_model(args):
            return globals()[args['model_type']]

        def train(args):
            m3, loss, opt = get_trainer(args)
            out = {'model': load_model(args).__module__, 'model3d': m3.__module__, 'loss': loss.__module__,
                   'optimizer': opt.__module__, 'collate': globals()[args['collate_function']].__module__,
                   'other': globals()['Other'].origin, 'argv': sys.argv[1:], 'name': __name__}
            print('RESULT ' + json.dumps(out))

        if __name__ == '__main__':
            train({'model_type': 'PNA', 'model3d_type': 'Net3D', 'loss_func': 'NTXent', 'optimizer': 'Adam',
                   'collate_function': 'contrastive_collate'})
    '''))
    before = script.read_text()
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'launch_reference.py'), str(script), '--config=x.yml'],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][0]
    import json
    out = json.loads(line[len('RESULT '):])
    assert out['model'] == '3dinfomax_amd.pna' and out['model3d'] == '3dinfomax_amd.net3d'
    assert out['loss'] == '3dinfomax_amd.losses' and out['collate'] == '3dinfomax_amd.graph'
    assert out['optimizer'] == '3dinfomax_amd.optim'
    assert out['other'] == 'reference'                       # names the plugin does not export stay the reference's
    assert out['argv'] == ['--config=x.yml'] and out['name'] == '__main__'
    assert script.read_text() == before
    # --keep leaves a reference class in place
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'launch_reference.py'), '--keep', 'Adam,NTXent', str(script)],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][0][len('RESULT '):])
    assert out['optimizer'].startswith('torch.optim') and out['loss'] == 'commons_losses' and out['model'] == '3dinfomax_amd.pna'
