#!/usr/bin/env python
"""Hyper-parameter search front end for the MI355X GRU4Rec path.

Takes the command line and the parameter-space files (`paramspaces/*.json`: one JSON object per line with `name`, `dtype`
int / float / categorical, `values`, optional `step` and `log`) of the reference's `paropt.py`, and maximises the same
quantity -- Recall@N or MRR@N of a model trained with `-fp` fixed parameters plus one sampled point.  Where the reference
starts `python run.py ...` once per trial and scrapes its `PRIMARY METRIC:` line, this keeps the event tables and the
device in the process: a trial costs its training time, not an interpreter start and a reload of the data.  Optuna drives
the sampling when it is installed (as in the reference); otherwise a seeded random search over the same space is used.
`--subprocess` restores the one-`run.py`-per-trial behaviour.
"""
import argparse
import importlib
import json
import math
import os
import random
import re
import subprocess
import sys
import time
from collections import OrderedDict

FLAGS = [
    (('path',), dict(metavar='PATH', help='training data (TAB separated file or pickled DataFrame)')),
    (('test',), dict(metavar='TEST_PATH', help='test data used to score every trial')),
    (('-g', '--gru4rec_model'), dict(metavar='GRFILE', default='gru4rec_amd.gru4rec', help='module providing the GRU4Rec class (default gru4rec_amd.gru4rec)')),
    (('-tf', '--theano_flags'), dict(metavar='FLAGS', nargs='?', default='', help='accepted for compatibility, unused')),
    (('-fp', '--fixed_parameters'), dict(metavar='PARAM_STRING', default='', help='parameters shared by all trials: name=value pairs separated by commas')),
    (('-opf', '--optuna_parameter_file'), dict(metavar='PATH', required=True, help='parameter space: one JSON object per line')),
    (('-m', '--measure'), dict(metavar='AT', type=int, nargs='?', default=20, help='cut-off of the optimised metric (default 20)')),
    (('-nt', '--ntrials'), dict(metavar='NT', type=int, nargs='?', default=50, help='number of trials (default 50)')),
    (('-fm', '--final_measure'), dict(metavar='AT', type=int, nargs='*', default=[20], help='cut-offs reported for the best point (default 20)')),
    (('-pm', '--primary_metric'), dict(metavar='METRIC', choices=['recall', 'mrr'], default='recall', help='recall or mrr (default recall)')),
    (('-e', '--eval_type'), dict(metavar='EVAL_TYPE', choices=['standard', 'conservative', 'median', 'tiebreaking'], default='standard', help='tie handling (default standard)')),
    (('-ik', '--item_key'), dict(metavar='IK', default='ItemId', help='item id column (default ItemId)')),
    (('-sk', '--session_key'), dict(metavar='SK', default='SessionId', help='session id column (default SessionId)')),
    (('-tk', '--time_key'), dict(metavar='TK', default='Time', help='timestamp column (default Time)')),
    (('--seed',), dict(type=int, default=0, help='seed of the built-in sampler (default 0)')),
    (('--sampler',), dict(choices=['auto', 'optuna', 'random'], default='auto', help='auto = optuna when importable, else random search')),
    (('--subprocess',), dict(action='store_true', help='run every trial as `python run.py ...` like the reference')),
]


class Dimension:
    """One line of a parameter-space file."""

    def __init__(self, spec):
        self.name, self.kind, self.values = spec['name'], spec['dtype'], spec['values']
        self.log = bool(spec.get('log', False))
        self.step = spec.get('step', 1 if self.kind == 'int' else None)
        if self.kind not in ('int', 'float', 'categorical'):
            raise ValueError('parameter {}: unknown dtype {}'.format(self.name, self.kind))
        if not isinstance(self.values, list) or (self.kind != 'categorical' and len(self.values) != 2):
            raise ValueError('parameter {}: `values` must be [low, high] (or the list of options)'.format(self.name))

    def describe(self):
        if self.kind == 'categorical':
            return 'PARAMETER {} \t type={} \t options: [{}]'.format(self.name, self.kind, ','.join(str(v) for v in self.values))
        return 'PARAMETER {} \t type={} \t range=[{}..{}] (step={}) \t {} scale'.format(
            self.name, self.kind, self.values[0], self.values[1], 'N/A' if self.step is None else self.step, 'LOG' if self.log else 'UNIFORM')

    def draw(self, rng):
        """Uniform (or log-uniform) draw on the grid optuna's suggest_int / suggest_float would use."""
        if self.kind == 'categorical':
            return rng.choice(self.values)
        lo, hi = (int(v) for v in self.values) if self.kind == 'int' else (float(v) for v in self.values)
        if self.log:
            x = math.exp(rng.uniform(math.log(lo), math.log(hi)))
            return min(hi, max(lo, int(round(x)))) if self.kind == 'int' else x
        if self.step is None:
            return rng.uniform(lo, hi)
        k = rng.randint(0, int(math.floor((hi - lo) / self.step + 1e-9)))
        x = lo + k * self.step
        return int(x) if self.kind == 'int' else round(x, 10)

    def suggest(self, trial):
        if self.kind == 'int':
            return trial.suggest_int(self.name, int(self.values[0]), int(self.values[1]), step=self.step, log=self.log)
        if self.kind == 'float':
            return trial.suggest_float(self.name, float(self.values[0]), float(self.values[1]), step=self.step, log=self.log)
        return trial.suggest_categorical(self.name, self.values)


def read_space(fname):
    with open(fname, 'rt') as fh:
        return [Dimension(json.loads(line)) for line in fh if line.strip()]


def param_string(point):
    return ','.join('{}={}'.format(k, v) for k, v in point.items())


class InProcessRunner:
    """Loads the tables once; a trial = set_params -> fit -> evaluate_gpu in this process."""

    def __init__(self, opts):
        import run as frontend
        self.opts, self.frontend = opts, frontend
        self.model_cls = importlib.import_module(opts.gru4rec_model).GRU4Rec
        print('Loading training data...')
        self.train = frontend.read_events(opts.path, opts, self.model_cls)
        print('Loading test data...')
        self.test = frontend.read_events(opts.test, opts, self.model_cls)
        self.evaluation = importlib.import_module('gru4rec_amd.evaluation')

    def __call__(self, point, cuts):
        opts = self.opts
        settings = OrderedDict(kv.split('=') for kv in opts.fixed_parameters.split(',') if kv)
        settings.update((k, str(v)) for k, v in point.items())
        model = self.model_cls()
        model.set_params(**settings)
        began = time.time()
        model.fit(self.train)
        print('Total training time: {:.2f}s'.format(time.time() - began))
        try:
            return self.evaluation.evaluate_gpu(model, self.test, batch_size=512, cut_off=list(cuts), mode=opts.eval_type,
                                                item_key=opts.item_key, session_key=opts.session_key, time_key=opts.time_key)
        finally:
            if hasattr(model, 'close'):
                model.close()


class SubprocessRunner:
    """The reference's way: one `python run.py` per trial, results scraped from its output."""

    def __init__(self, opts):
        self.opts = opts

    def __call__(self, point, cuts):
        o = self.opts
        both = ','.join(s for s in (o.fixed_parameters, param_string(point)) if s)
        cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'run.py'), o.path, '-t', o.test, '-g', o.gru4rec_model,
               '-ps', both, '-m'] + [str(c) for c in cuts] + ['-e', o.eval_type, '-ik', o.item_key, '-sk', o.session_key, '-tk', o.time_key]
        recall, mrr = [], []
        with subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) as proc:
            for line in proc.stdout:
                line = line.rstrip()
                print(line)
                hit = re.match(r'Recall@\d+: (\S+) MRR@\d+: (\S+)', line)
                if hit:
                    recall.append(float(hit.group(1)))
                    mrr.append(float(hit.group(2)))
        if len(recall) != len(cuts):
            raise RuntimeError('run.py did not report the metrics (exit code {})'.format(proc.returncode))
        return recall, mrr


def search(space, objective, n_trials, sampler, seed):
    """Returns (best value, best point)."""
    use_optuna = False
    if sampler in ('auto', 'optuna'):
        try:
            import optuna
            use_optuna = True
        except ImportError:
            if sampler == 'optuna':
                raise
    if use_optuna:
        study = optuna.create_study(direction='maximize')
        study.optimize(lambda trial: objective(OrderedDict((d.name, d.suggest(trial)) for d in space)), n_trials=n_trials)
        return study.best_value, OrderedDict(study.best_params)
    rng = random.Random(seed)
    best = (-math.inf, None)
    for number in range(n_trials):
        point = OrderedDict((d.name, d.draw(rng)) for d in space)
        value = objective(point)
        if value > best[0]:
            best = (value, point)
        print('Trial {} finished with value: {} and parameters: {}. Best is {} with value: {}.'.format(
            number, value, dict(point), dict(best[1]), best[0]))
    return best


def main(argv=None):
    ap = argparse.ArgumentParser(description='Search GRU4Rec hyper-parameters on an MI355X (recall / MRR on the given test set).')
    for names, kw in FLAGS:
        ap.add_argument(*names, **kw)
    opts = ap.parse_args(argv)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    space = read_space(opts.optuna_parameter_file)
    print('-' * 80)
    print('PARAMETER SPACE')
    for dim in space:
        print('\t' + dim.describe())
    print('-' * 80)
    runner = SubprocessRunner(opts) if opts.subprocess else InProcessRunner(opts)
    which = ('recall', 'mrr').index(opts.primary_metric)

    def objective(point):
        print('Trial parameters: {}'.format(param_string(point)))
        value = runner(point, [opts.measure])[which][0]
        print('PRIMARY METRIC: {}'.format(value))
        return value

    best_value, best_point = search(space, objective, opts.ntrials, opts.sampler, opts.seed)
    print('Best {}@{}: {} with {}'.format(opts.primary_metric, opts.measure, best_value, param_string(best_point)))
    print('Running final eval @{}:'.format(opts.final_measure))
    recall, mrr = runner(best_point, opts.final_measure)
    for cut, r, m in zip(opts.final_measure, recall, mrr):
        print('Recall@{}: {:.6f} MRR@{}: {:.6f}'.format(cut, r, cut, m))
    return best_value, best_point


if __name__ == '__main__':
    main()
