#!/usr/bin/env python
"""Command line front end of the MI355X GRU4Rec path.

It accepts the same flags, prints the same progress lines and uses the same exit codes as the reference's `run.py`, so
shell scripts and `paropt.py` (which scrapes the `PRIMARY METRIC:` line) keep working; the work itself goes through the
plugin class selected with `-g` (any module exposing `GRU4Rec`; default: the HIP implementation).
"""
import argparse
import importlib
import importlib.util
import os
import sys
import time
from collections import OrderedDict

KEY_COLUMNS = (          # (what it is, CLI attribute, default column, model parameter that overrides it)
    ('session IDs', 'session_key', 'SessionId'),
    ('item IDs', 'item_key', 'ItemId'),
    ('time', 'time_key', 'Time'),
)

# flag table: (short, long, argparse keywords)
OPTIONS = [
    ('-ps', '--parameter_string', dict(metavar='PARAM_STRING', help='hyper-parameters inline: name=value pairs separated by commas; list values use "/" (layers=100/100)')),
    ('-pf', '--parameter_file', dict(metavar='PARAM_PATH', help='python file that defines an OrderedDict called gru4rec_params')),
    ('-l', '--load_model', dict(action='store_true', help='PATH is a saved model: skip training')),
    ('-s', '--save_model', dict(metavar='MODEL_PATH', help='where to pickle the trained model (not saved by default)')),
    ('-t', '--test', dict(metavar='TEST_PATH', nargs='+', help='one or more test sets to evaluate on')),
    ('-m', '--measure', dict(metavar='AT', type=int, nargs='+', default=[20], help='cut-offs N for Recall@N / MRR@N (default 20)')),
    ('-e', '--eval_type', dict(metavar='EVAL_TYPE', choices=['standard', 'conservative', 'median', 'tiebreaking'], default='standard',
                               help='tie handling when ranking the target item (default standard)')),
    ('-ss', '--sample_store_size', dict(metavar='SS', type=int, default=10000000, help='number of pre-drawn negative samples kept on the device (default 10000000)')),
    (None, '--sample_store_on_cpu', dict(action='store_true', help='draw the negative samples with the reference\'s host sampler (NumPy stream) and upload them store by store, instead of sampling on the device')),
    ('-g', '--gru4rec_model', dict(metavar='GRFILE', default='gru4rec_amd.gru4rec', help='module that provides the GRU4Rec class (default gru4rec_amd.gru4rec)')),
    ('-ik', '--item_key', dict(metavar='IK', default='ItemId', help='item id column (default ItemId)')),
    ('-sk', '--session_key', dict(metavar='SK', default='SessionId', help='session id column (default SessionId)')),
    ('-tk', '--time_key', dict(metavar='TK', default='Time', help='timestamp column (default Time)')),
    ('-pm', '--primary_metric', dict(metavar='METRIC', choices=['recall', 'mrr'], default='recall', help='metric reported on the PRIMARY METRIC line (default recall)')),
    ('-lpm', '--log_primary_metric', dict(action='store_true', help='print the PRIMARY METRIC line after every evaluation')),
    (None, '--sparse_exact', dict(action='store_true', help='with --gpus N: keep the replicas bit-identical by exchanging every rank\'s per-occurrence gradient rows of the item tables every step (one RCCL all-gather) instead of reconciling GPU-local rows every sync_every steps.  REDUCE form: all ranks draw ONE stream of negatives, the gradient rows of a shared negative are summed over the ranks before the optimizer rule and every row is scaled by 1 / N -- the update of one batch of N x batch_size rows, except that a row meets only its own rank\'s in-batch negatives; for catalogues whose exchanged list fits the LDS')),
    (None, '--gpus', dict(metavar='N', type=int, default=1, help='train on N GPUs of this node (not in the reference): one process per GPU, sessions sharded over '
                          'the ranks, dense GRU gradients all-reduced by RCCL every step, item rows GPU-local and reconciled every sync_every steps (4 at two ranks, 16 from three on) and at every epoch end; rank 0 saves / evaluates')),
]


def build_parser():
    ap = argparse.ArgumentParser(description='Train a GRU4Rec model on an MI355X (or load one) and report Recall@N / MRR@N.')
    ap.add_argument('path', metavar='PATH', help='training data (.tsv / .txt with TAB separators, or a pickled DataFrame) -- or the model file when -l is given')
    for short, long_, kw in OPTIONS:
        ap.add_argument(*([short, long_] if short else [long_]), **kw)
    return ap


def fail_missing_column(what, column, default, fname):
    print('ERROR. The column specified for {} "{}" is not in the data file ({})'.format(what, column, fname))
    print('The default column name is "{}", but you can specify otherwise by setting the `{}` parameter of the model.'.format(
        default, {'session IDs': 'session_key', 'item IDs': 'item_key', 'time': 'time_key'}[what]))
    sys.exit(1)


def read_events(fname, opts, model_cls=None):
    """Event table with the three key columns.  Item ids are kept as strings, session ids as int32.  Plugins that declare
    `accepts_categorical_items` get the table from the native multi-threaded parser (item ids as a categorical column)."""
    import pandas as pd
    wanted = [(what, getattr(opts, attr), default) for what, attr, default in KEY_COLUMNS]
    pickled = fname.endswith('.pickle')
    if pickled:
        import joblib
        print('Loading data from pickle file: {}'.format(fname))
        table = joblib.load(fname)
        present = set(table.columns)
    else:
        with open(fname, 'rt') as fh:
            present = set(fh.readline().rstrip('\r\n').split('\t'))
    for what, column, default in wanted:
        if column not in present:
            fail_missing_column(what, column, default, fname)
    if pickled:
        return table
    print('Loading data from TAB separated file: {}'.format(fname))
    if getattr(model_cls, 'accepts_categorical_items', False):
        from gru4rec_amd import eventio
        return eventio.read_events(fname, opts.session_key, opts.item_key, opts.time_key)
    cols = [c for _, c, _ in wanted]
    return pd.read_csv(fname, sep='\t', usecols=cols, dtype={opts.session_key: 'int32', opts.item_key: 'str'})


def hyper_parameters(opts):
    if opts.parameter_file:
        where = os.path.abspath(opts.parameter_file)
        spec = importlib.util.spec_from_file_location(os.path.splitext(os.path.basename(where))[0], where)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        print('Loaded parameters from file: {}'.format(where))
        return module.gru4rec_params
    return OrderedDict(item.split('=') for item in opts.parameter_string.split(','))


def train(model_cls, opts):
    params = hyper_parameters(opts)
    print('Creating GRU4Rec model')
    model = model_cls()
    model.set_params(**params)
    rank, world = 0, 1
    if opts.gpus > 1 or os.environ.get('G4R_FORCE_STAGED'):
        # a rank of `run.py --gpus N` (spawned by main() below, or by torch.distributed.run): the communicator is created inside fit
        from gru4rec_amd import launch
        rank, world, local_rank = launch.layout()
        if world != opts.gpus:
            print('ERROR. WORLD_SIZE ({}) does not match --gpus ({})'.format(world, opts.gpus))
            sys.exit(1)
        if not hasattr(model, 'set_distributed'):
            print('ERROR. The model class {} does not support --gpus'.format(model_cls.__module__))
            sys.exit(1)
        model.device = local_rank
        model.sparse_exact = bool(opts.sparse_exact)
        model.set_distributed(rank, world, launch.unique_id(rank, world))
    print('Loading training data...')
    events = read_events(opts.path, opts, model_cls)
    if opts.sample_store_on_cpu:
        print('WARNING! The sample store is set to be on the CPU. This will make training significantly slower on the GPU.')
    print('Started training')
    started = time.time()
    model.fit(events, sample_store=opts.sample_store_size, store_type='cpu' if opts.sample_store_on_cpu else 'gpu')
    print('Total training time: {:.2f}s'.format(time.time() - started))
    if rank != 0:
        # replicas are identical after the last reconciliation: rank 0 alone saves and evaluates
        model.close()
        sys.exit(0)
    if world > 1:
        from gru4rec_amd import launch
        launch.cleanup(rank)
    if opts.save_model is not None:
        print('Saving trained model to: {}'.format(opts.save_model))
        model.savemodel(opts.save_model)
    return model


def evaluate(model, opts):
    from gru4rec_amd import evaluation
    which = ('recall', 'mrr').index(opts.primary_metric.lower())
    for fname in opts.test:
        print('Loading test data...')
        events = read_events(fname, opts, type(model))
        print('Starting evaluation (cut-off={}, using {} mode for tiebreaking)'.format(opts.measure, opts.eval_type))
        started = time.time()
        scores = evaluation.evaluate_gpu(model, events, batch_size=512, cut_off=opts.measure, mode=opts.eval_type,
                                         item_key=opts.item_key, session_key=opts.session_key, time_key=opts.time_key)
        print('Evaluation took {:.2f}s'.format(time.time() - started))
        for pos, cut in enumerate(opts.measure):
            print('Recall@{}: {:.6f} MRR@{}: {:.6f}'.format(cut, scores[0][pos], cut, scores[1][pos]))
        if opts.log_primary_metric:
            print('PRIMARY METRIC: {}'.format(scores[which][0]))


def main(argv=None):
    opts = build_parser().parse_args(argv)
    sources = [opts.parameter_string is not None, opts.parameter_file is not None, bool(opts.load_model)]
    if sum(sources) != 1:
        print('ERROR. Exactly one of the following parameters must be provided: --parameter_string, --parameter_file, --load_model')
        sys.exit(1)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if opts.gpus > 1 and 'WORLD_SIZE' not in os.environ and not opts.load_model:
        # become the launcher: N ranks of this very command line (rank 0 keeps stdout, the others' progress lines go to stderr)
        from gru4rec_amd import launch
        sys.exit(launch.spawn(__file__, sys.argv[1:] if argv is None else list(argv), opts.gpus))
    model_cls = importlib.import_module(opts.gru4rec_model).GRU4Rec
    if opts.load_model:
        print('Loading trained model from file: {}'.format(opts.path))
        model = model_cls.loadmodel(opts.path)
    else:
        model = train(model_cls, opts)
    if opts.test is not None:
        evaluate(model, opts)


if __name__ == '__main__':
    main()
