"""Host-side mirror of the reference's class / CLI surface (no GPU needed): constructor defaults, set_params
coercion and messages (gru4rec.py:162-187), error conventions, parameter files, synthetic data generator."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from gru4rec_amd import _native, synth
from gru4rec_amd.gru4rec import GRU4Rec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constructor_defaults_match_reference():
    g = GRU4Rec()
    assert (g.loss, g.final_act, g.hidden_act, g.layers, g.n_epochs, g.batch_size) == ('bpr-max', 'linear', 'tanh', [100], 10, 32)
    assert (g.learning_rate, g.momentum, g.n_sample, g.sample_alpha, g.bpreg, g.logq) == (0.1, 0.0, 2048, 0.75, 1.0, 0.0)
    assert (g.constrained_embedding, g.embedding, g.adapt, g.time_sort, g.train_random_order) == (False, 0, 'adagrad', True, False)
    assert (g.session_key, g.item_key, g.time_key) == ('SessionId', 'ItemId', 'Time')


def test_set_params_string_coercion(capsys):
    g = GRU4Rec()
    g.set_params(loss='cross-entropy', layers='64/32', constrained_embedding='True', learning_rate='0.05',
                 final_act='elu-0.5', n_sample='128', embedding='layersize')
    assert g.layers == [64, 32] and g.constrained_embedding is True and g.learning_rate == 0.05
    assert g.n_sample == 128 and g.embedding == 64 and g.loss == 'cross-entropy'
    out = capsys.readouterr().out
    assert 'SET   loss' in out and "TO   cross-entropy" in out and "(type: <class 'str'>)" in out


def test_unknown_names_raise_notimplemented(capsys):
    g = GRU4Rec()
    with pytest.raises(NotImplementedError):
        g.set_params(no_such_param='1')
    assert 'Unkown attribute: no_such_param' in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        g.set_params(constrained_embedding='maybe')
    with pytest.raises(NotImplementedError):
        GRU4Rec(loss='hinge')
    with pytest.raises(NotImplementedError):
        GRU4Rec(final_act='gelu')
    with pytest.raises(NotImplementedError):
        GRU4Rec(hidden_act='softmax')


def test_reference_paramfiles_load(tmp_path):
    """The OrderedDict parameter files of the reference (paramfiles/*.py shape) go through set_params unchanged."""
    from collections import OrderedDict
    params = OrderedDict([('loss', 'cross-entropy'), ('constrained_embedding', True), ('embedding', 0),
                          ('final_act', 'softmax'), ('layers', [100]), ('n_epochs', 10), ('batch_size', 32),
                          ('dropout_p_embed', 0.0), ('dropout_p_hidden', 0.4), ('learning_rate', 0.2), ('momentum', 0.2),
                          ('n_sample', 2048), ('sample_alpha', 0.5), ('bpreg', 0.0), ('logq', 1.0)])
    g = GRU4Rec()
    g.set_params(**params)
    assert g.logq == 1.0 and g.dropout_p_hidden == 0.4 and g._loss_id == _native.LOSS_IDS['cross-entropy']


def test_out_of_scope_options_fail_loudly_at_fit():
    data = synth.make_sessions(50, n_items=30, seed=1)
    for kw in (dict(smoothing=0.1, loss='bpr-max', constrained_embedding=True),
               dict(layers=[400])):   # last: one-hot input wider than the 1024-float row limit (3 x 400)
        kw.setdefault('layers', [8])
        g = GRU4Rec(batch_size=4, **kw)
        g.n_items = 30
        with pytest.raises(NotImplementedError):
            g._check_supported()


@pytest.mark.parametrize('params,needle', [
    (dict(layers='1100', constrained_embedding='True'), 'up to 1024 units'),
    (dict(embedding='2000', constrained_embedding='False'), 'embeddings of up to 1024'),
    (dict(layers='400', constrained_embedding='False', embedding='0'), 'layers[0] <= 340'),
    (dict(batch_size='512', n_sample='40000'), 'about 38,000 rows / columns')])
def test_shape_limits_are_refused_at_set_params_time_with_the_limit_in_the_message(params, needle):
    """The reference has no such limits; the MI355X path names its own where the configuration is made (run.py -> set_params), in
    the reference's way of refusing a configuration (NotImplementedError, gru4rec.py:143-177)."""
    g = GRU4Rec()
    with pytest.raises(NotImplementedError) as e:
        g.set_params(**params)
    assert needle in str(e.value)


def test_shapes_inside_the_limits_pass_set_params():
    GRU4Rec().set_params(layers='1024', constrained_embedding='True', batch_size='512', n_sample='29000')
    GRU4Rec().set_params(layers='100', constrained_embedding='True', batch_size='512', n_sample='36000')
    GRU4Rec().set_params(layers='340', constrained_embedding='False', embedding='0', n_sample='2048')
    GRU4Rec().set_params(layers='96/1024', embedding='1000')


def test_unknown_adapt_means_plain_sgd_like_the_reference():
    """gru4rec.py:392-399,411-418: any `adapt` other than the four known names falls through to the unscaled gradient."""
    g = GRU4Rec(layers=[8], batch_size=4, adapt='sgd', constrained_embedding=True)
    g._check_supported()
    assert _native.ADAPT_IDS.get(g.adapt, _native.ADAPT_IDS[None]) == _native.ADAPT_IDS[None]


def test_fit_without_gpu_raises_not_falls_back():
    if _native.device_count() > 0:
        pytest.skip('GPU present')
    data = synth.make_sessions(60, n_items=30, seed=1)
    g = GRU4Rec(layers=[8], batch_size=4, n_sample=8, constrained_embedding=True)
    with pytest.raises(_native.NativeError):
        g.fit(data)


def test_run_py_argument_errors():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'x.tsv'], capture_output=True, text=True)
    assert r.returncode == 1 and 'Exactly one of the following parameters must be provided' in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'x.tsv', '-ps', 'loss=bpr-max', '-l'],
                       capture_output=True, text=True)
    assert r.returncode == 1


def test_synthetic_generator_is_rsc15_shaped_and_deterministic():
    a = synth.make_sessions(4000, n_items=2000, seed=3)
    b = synth.make_sessions(4000, n_items=2000, seed=3)
    pd.testing.assert_frame_equal(a, b)
    lens = a.groupby('SessionId').size()
    assert lens.min() >= 2 and 3.0 < lens.mean() < 5.0
    assert (a.sort_values(['SessionId', 'Time']).index == a.index).all()
    train, test = synth.train_test_split(a)
    assert len(test) > 0 and set(test.ItemId) <= set(train.ItemId)
    assert test.groupby('SessionId').size().min() >= 2
