"""Zero-edit launcher: run the reference's own `train.py` with the MI355X plugin classes bound in - the file stays
untouched.

    cd /path/to/3DInfomax
    python /path/to/this/repo/launch_reference.py train.py --config=configs_clean/pre-train_QM9.yml

How: the reference resolves its model, 3D model, loss, optimizer and collate function by NAME at call time through
`globals()[...]` of the `train` module (reference train.py:167-172, 189, 208-209, 589-590), after a block of star-imports
(train.py:50-56).  This launcher (1) executes `train.py` as the module `train` - its `if __name__ == '__main__':` block
does not run, (2) rebinds the plugin names in that module's namespace (and in `trainer.trainer`, whose checkpoint code
looks the model class up the same way, trainer/trainer.py:266-270), (3) executes the body of the `__main__` block -
taken from the file's own syntax tree - inside the `train` namespace with `sys.argv` set to what `train.py` would see.

`--keep NAME[,NAME...]` (before the script path) leaves reference classes in place, e.g. `--keep Adam`.
"""
import ast
import importlib
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def plugin_names(keep=()):
    """{name: object} of everything the drop-in rebinds (the package's __all__ minus `keep`)."""
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    pkg = importlib.import_module('3dinfomax_amd')
    return {n: getattr(pkg, n) for n in pkg.__all__ if n not in keep}


def rebind(namespace, names):
    """bind `names` into a module namespace; returns the names that replaced an existing binding"""
    replaced = [n for n in names if n in namespace and namespace[n] is not names[n]]
    namespace.update(names)
    return replaced


def load_script_as_module(path, name='train'):
    """execute `path` as module `name` (so that its __main__ block stays dormant) and return (module, source)"""
    path = os.path.abspath(path)
    script_dir = os.path.dirname(path)
    if script_dir not in sys.path:
        sys.path.insert(0, script_dir)          # what `python train.py` does: the script's directory is importable
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    with open(path) as f:
        return mod, f.read()


def main_block(source, filename):
    """code object of the body of the top-level `if __name__ == '__main__':` statement(s) of `source`"""
    tree = ast.parse(source, filename)
    body = []
    for node in tree.body:
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name) \
                and node.test.left.id == '__name__' and len(node.test.comparators) == 1 \
                and isinstance(node.test.comparators[0], ast.Constant) and node.test.comparators[0].value == '__main__':
            body.extend(node.body)
    if not body:
        raise SystemExit(f'{filename}: no `if __name__ == "__main__":` block to run')
    return compile(ast.Module(body=body, type_ignores=[]), filename, 'exec')


def run(script, argv, keep=(), extra_namespaces=('trainer.trainer',)):
    names = plugin_names(keep)
    mod, source = load_script_as_module(script)
    replaced = rebind(mod.__dict__, names)
    for ns in extra_namespaces:
        m = sys.modules.get(ns)
        if m is not None:
            rebind(m.__dict__, {k: v for k, v in names.items() if k in m.__dict__})
    print(f'[launch_reference] {os.path.basename(script)}: bound {len(names)} MI355X plugin names '
          f'({len(replaced)} replaced reference bindings: {", ".join(sorted(replaced)) or "-"})', file=sys.stderr)
    sys.argv = [script] + list(argv)
    code = main_block(source, os.path.abspath(script))
    mod.__dict__['__name__'] = '__main__'       # code inside the block may test it again
    exec(code, mod.__dict__)
    return mod


if __name__ == '__main__':
    args = sys.argv[1:]
    keep = ()
    if args and args[0] == '--keep':
        keep = tuple(args[1].split(','))
        args = args[2:]
    elif args and args[0].startswith('--keep='):
        keep = tuple(args[0].split('=', 1)[1].split(','))
        args = args[1:]
    if not args:
        raise SystemExit(__doc__)
    run(args[0], args[1:], keep)
