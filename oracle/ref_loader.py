"""TEST INFRASTRUCTURE.  Loads the REFERENCE's own modules (/root/reference/gru4rec.py, evaluation.py) by explicit file
path on top of the Theano stand-in (oracle/theano_shim), under their own module names -- `gru4rec`, `evaluation` --
so that pickles written by the reference (`gru4rec.GRU4Rec`, gru4rec.py:742-756) resolve.  Nothing in the product
imports this; it only works where /root/reference exists (golden / fixture generation, optional CPU tests)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'


def available():
    return os.path.exists(os.path.join(REF, 'gru4rec.py'))


def load():
    """Returns (theano_shim, reference gru4rec module, reference evaluation module); reloads on every call."""
    shim = os.path.join(HERE, 'theano_shim')
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p) if p == shim else sys.path.append(p)
    if ROOT not in sys.path:
        sys.path.insert(1, ROOT)
    import theano                      # the stand-in
    mods = []
    for name in ('gru4rec', 'evaluation'):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod        # the reference's pickles name this module
        spec.loader.exec_module(mod)
        mods.append(mod)
    return theano, mods[0], mods[1]


def unload():
    for name in ('gru4rec', 'evaluation'):
        sys.modules.pop(name, None)
