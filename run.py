#!/usr/bin/env python
"""CLI with the arguments, messages and exit codes of the reference's run.py (run.py:10-133), driving the
MI355X implementation.  `-g/--gru4rec_model` (run.py:21,39) keeps working as the plugin hook: any module that
exposes a `GRU4Rec` class; the default is gru4rec_amd.gru4rec.  `paropt.py` of the reference works unchanged
against this script (it scrapes the `PRIMARY METRIC:` line).
"""
import argparse
import importlib
import importlib.util
import os
import shutil
import sys
import time
from collections import OrderedDict


class _WideHelp(argparse.HelpFormatter):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._width = shutil.get_terminal_size().columns


def build_parser():
    p = argparse.ArgumentParser(formatter_class=_WideHelp, description='Train or load a GRU4Rec model & measure recall and MRR on the specified test set(s).')
    p.add_argument('path', metavar='PATH', type=str, help='Path to the training data (TAB separated file (.tsv or .txt) or pickled pandas.DataFrame object (.pickle)) (if the --load_model parameter is NOT provided) or to the serialized model (if the --load_model parameter is provided).')
    p.add_argument('-ps', '--parameter_string', metavar='PARAM_STRING', type=str, help='Training parameters as `name1=value1,name2=value2...`; lists use / (e.g. layers=200/200).')
    p.add_argument('-pf', '--parameter_file', metavar='PARAM_PATH', type=str, help='Config file containing a single OrderedDict named `gru4rec_params`.')
    p.add_argument('-l', '--load_model', action='store_true', help='Load an already trained model instead of training a model.')
    p.add_argument('-s', '--save_model', metavar='MODEL_PATH', type=str, help='Save the trained model to the MODEL_PATH. (Default: don\'t save model)')
    p.add_argument('-t', '--test', metavar='TEST_PATH', type=str, nargs='+', help='Path to the test data set(s) located at TEST_PATH.')
    p.add_argument('-m', '--measure', metavar='AT', type=int, nargs='+', default=[20], help='Measure recall & MRR at the defined recommendation list length(s). (Default: 20)')
    p.add_argument('-e', '--eval_type', metavar='EVAL_TYPE', choices=['standard', 'conservative', 'median', 'tiebreaking'], default='standard', help='How ties between prediction scores are handled. (Default: standard)')
    p.add_argument('-ss', '--sample_store_size', metavar='SS', type=int, default=10000000, help='Size of the negative sample buffer. (Default: 10000000)')
    p.add_argument('--sample_store_on_cpu', action='store_true', help='Kept for CLI compatibility: the MI355X path always keeps the sample store in HBM.')
    p.add_argument('-g', '--gru4rec_model', metavar='GRFILE', type=str, default='gru4rec_amd.gru4rec', help='Module containing the GRU4Rec class. (Default: gru4rec_amd.gru4rec)')
    p.add_argument('-ik', '--item_key', metavar='IK', type=str, default='ItemId', help='Column name corresponding to the item IDs (detault: ItemId).')
    p.add_argument('-sk', '--session_key', metavar='SK', type=str, default='SessionId', help='Column name corresponding to the session IDs (default: SessionId).')
    p.add_argument('-tk', '--time_key', metavar='TK', type=str, default='Time', help='Column name corresponding to the timestamp (default: Time).')
    p.add_argument('-pm', '--primary_metric', metavar='METRIC', choices=['recall', 'mrr'], default='recall', help='Set primary metric, recall or mrr (e.g. for paropt). (Default: recall)')
    p.add_argument('-lpm', '--log_primary_metric', action='store_true', help='If provided, evaluation will log the value of the primary metric at the end of the run.')
    return p


def _column_error(kind, key, fname):
    print('ERROR. The column specified for {} "{}" is not in the data file ({})'.format(kind, key, fname))
    default = {'session IDs': ('SessionId', 'session_key'), 'item IDs': ('ItemId', 'item_key'), 'time': ('Time', 'time_key')}[kind]
    print('The default column name is "{}", but you can specify otherwise by setting the `{}` parameter of the model.'.format(*default))
    sys.exit(1)


def load_data(fname, args):
    """TSV or pickled DataFrame with the three key columns; ItemId is read as str like the reference (run.py:77)."""
    import joblib
    import pandas as pd
    keys = (('session IDs', args.session_key), ('item IDs', args.item_key), ('time', args.time_key))
    if fname.endswith('.pickle'):
        print('Loading data from pickle file: {}'.format(fname))
        data = joblib.load(fname)
        for kind, key in keys:
            if key not in data.columns:
                _column_error(kind, key, fname)
        return data
    with open(fname, 'rt') as f:
        header = f.readline().strip().split('\t')
    for kind, key in keys:
        if key not in header:
            _column_error(kind, key, fname)
    print('Loading data from TAB separated file: {}'.format(fname))
    return pd.read_csv(fname, sep='\t', usecols=[args.session_key, args.item_key, args.time_key],
                       dtype={args.session_key: 'int32', args.item_key: 'str'})


def main(argv=None):
    args = build_parser().parse_args(argv)
    if (args.parameter_string is not None) + (args.parameter_file is not None) + (args.load_model) != 1:
        print('ERROR. Exactly one of the following parameters must be provided: --parameter_string, --parameter_file, --load_model')
        sys.exit(1)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    GRU4Rec = importlib.import_module(args.gru4rec_model).GRU4Rec
    from gru4rec_amd import evaluation
    if args.load_model:
        print('Loading trained model from file: {}'.format(args.path))
        gru = GRU4Rec.loadmodel(args.path)
    else:
        if args.parameter_file:
            param_file_path = os.path.abspath(args.parameter_file)
            spec = importlib.util.spec_from_file_location(os.path.basename(param_file_path).split('.py')[0], param_file_path)
            params = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(params)
            gru4rec_params = params.gru4rec_params
            print('Loaded parameters from file: {}'.format(param_file_path))
        if args.parameter_string:
            gru4rec_params = OrderedDict([x.split('=') for x in args.parameter_string.split(',')])
        print('Creating GRU4Rec model')
        gru = GRU4Rec()
        gru.set_params(**gru4rec_params)
        print('Loading training data...')
        data = load_data(args.path, args)
        store_type = 'cpu' if args.sample_store_on_cpu else 'gpu'
        if store_type == 'cpu':
            print('WARNING! The sample store is set to be on the CPU. This will make training significantly slower on the GPU.')
        print('Started training')
        t0 = time.time()
        gru.fit(data, sample_store=args.sample_store_size, store_type=store_type)
        print('Total training time: {:.2f}s'.format(time.time() - t0))
        if args.save_model is not None:
            print('Saving trained model to: {}'.format(args.save_model))
            gru.savemodel(args.save_model)
    if args.test is not None:
        pm_index = {'recall': 0, 'mrr': 1}[args.primary_metric.lower()]
        for test_file in args.test:
            print('Loading test data...')
            test_data = load_data(test_file, args)
            print('Starting evaluation (cut-off={}, using {} mode for tiebreaking)'.format(args.measure, args.eval_type))
            t0 = time.time()
            res = evaluation.evaluate_gpu(gru, test_data, batch_size=512, cut_off=args.measure, mode=args.eval_type,
                                          item_key=args.item_key, session_key=args.session_key, time_key=args.time_key)
            print('Evaluation took {:.2f}s'.format(time.time() - t0))
            for i, c in enumerate(args.measure):
                print('Recall@{}: {:.6f} MRR@{}: {:.6f}'.format(c, res[0][i], c, res[1][i]))
            if args.log_primary_metric:
                print('PRIMARY METRIC: {}'.format(res[pm_index][0]))


if __name__ == '__main__':
    main()
