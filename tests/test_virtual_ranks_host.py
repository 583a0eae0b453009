"""Host control flow of gru4rec_amd/virtual_ranks.py without a GPU: the epoch loop must terminate and cut its chunks at the
reconciliation points for every combination of rank count / sync_every / chunk (an earlier version looped forever on one rank with
sync_every set: `n = min(n, sync_every - since)` reached 0).  The device model and the lock-step call are replaced by recorders."""
import numpy as np
import pytest

from gru4rec_amd import synth, virtual_ranks


class FakeModel:
    def __init__(self):
        self.calls, self.T = [], 0
        self.cfg = type('cfg', (), dict(batch_size=8))()

    def set_plan(self, plan):
        self.T = int(plan['T'])

    def reset_hidden(self):
        pass

    def sync_enable(self):
        pass

    def sync_set_rule(self, *a):
        pass

    def train_steps(self, t0, n):
        assert n >= 1 and t0 + n <= self.T
        self.calls.append((t0, n))

    def get_losses(self, t0, n):
        return np.zeros(n, dtype=np.float32)

    def sync_export(self, g):
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32)

    def sync_import(self, parts, g):
        pass

    def close(self):
        pass


@pytest.mark.parametrize('nranks,sync_every,chunk', [(1, 'default', 64), (1, 4, 7), (2, 'default', 64), (2, 4, 7), (3, None, 5), (2, 5, 3)])
def test_epoch_loop_terminates_and_cuts_at_the_reconciliation_points(monkeypatch, nranks, sync_every, chunk):
    from gru4rec_amd.gru4rec import GRU4Rec
    recon = []

    def fake_prepare(self, data, sample_store=0, store_type='gpu', resume=False):
        d = data.sort_values(['SessionId', 'Time'])
        self.n_items = int(d.ItemId.nunique())
        sizes = d.groupby('SessionId').size().values
        self._offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        self._base_order = np.arange(len(sizes))
        self._data_items = np.unique(d.ItemId.values, return_inverse=True)[1].astype(np.int32)
        self._model = FakeModel()

    def fake_build(offsets, order, items, B, ns, rank=0, nranks=1):
        T = 23 + rank        # ranks hold plans of different lengths: padded to the longest
        return dict(in_idx=np.zeros((T, B), np.int32), out_idx=np.zeros((T, B), np.int32), reset=np.zeros((T, B), np.uint8),
                    M=np.full(T, B, np.int32), T=T, n_compact=0, compact_steps=np.zeros(0, np.int64), compact_maps=np.zeros((0, B), np.int32))

    def fake_lockstep(models, t0, n):
        for m in models:
            m.train_steps(t0, n)

    monkeypatch.setattr(GRU4Rec, 'prepare', fake_prepare)
    monkeypatch.setattr(GRU4Rec, '_download_weights', lambda self: None)
    monkeypatch.setattr(virtual_ranks, 'build_rank_plan', fake_build)
    monkeypatch.setattr(virtual_ranks._native, 'virtual_train_steps', fake_lockstep)
    monkeypatch.setattr(virtual_ranks, 'reconcile', lambda models, groups=(0,): recon.append(models[0].calls[-1]) or 0)
    data = synth.make_sessions(40, n_items=20, seed=1)
    grus, st = virtual_ranks.fit_virtual_ranks(dict(layers=[8], batch_size=8, n_sample=0, n_epochs=2, constrained_embedding=True), data, nranks,
                                               sync_every=sync_every, chunk=chunk)
    T = 23 + nranks - 1
    assert st['steps'] == [T, T]
    for g in grus:
        calls = g._model.calls
        assert sum(n for _, n in calls) == 2 * T                      # every step of both epochs, once
        per_epoch = [c for c in calls[:len(calls) // 2]]
        assert [t for t, _ in per_epoch] == list(np.cumsum([0] + [n for _, n in per_epoch[:-1]]))      # contiguous
    k = grus[0].sync_steps(nranks) if sync_every == 'default' else sync_every
    if sync_every == 'default':
        assert k == (4 if nranks == 2 else 16)      # 'auto': two ranks need the tighter coupling
    if nranks > 1:
        expect = (((T - 1) // k) if k else 0) + 1                        # every k steps inside the epoch, and at its end
        assert st['syncs'] == 2 * expect
        if k:
            assert all((t + n) % k == 0 or t + n == T for t, n in recon)
    else:
        assert st['syncs'] == 0


@pytest.mark.parametrize('dev', [False, True])
@pytest.mark.parametrize('sync_every,steps_per_call,T', [(16, 16384, 50), (4, 7, 23), (0, 10, 23), (16, 5, 16), ('auto', 16384, 23)])
def test_run_epoch_reconciles_every_sync_every_steps_on_every_rank(sync_every, steps_per_call, T, dev):
    """GRU4Rec.run_epoch with a (fake) communicator: C-ABI calls are cut at the reconciliation points, g4r_comm_sync_sparse runs every
    `sync_every` steps (never behind the last step: fit() reconciles at the epoch end itself), the NaN exits stay collective."""
    from gru4rec_amd.gru4rec import GRU4Rec

    class Comm(FakeModel):
        def __init__(self):
            super().__init__()
            self.syncs, self.maxes = [], 0

        def comm_max(self, v):
            self.maxes += 1
            return v

        def comm_sync_sparse(self):
            self.syncs.append(sum(n for _, n in self.calls))

        def set_sync_every(self, k):
            self.dev_k = k
            return dev

    g = GRU4Rec(layers=[8], batch_size=8, n_sample=0, constrained_embedding=True)
    g.sync_every, g.steps_per_call = sync_every, steps_per_call
    g._dist = dict(rank=0, nranks=2, unique_id=b'x')
    g._model = Comm()
    g._model.T = T
    plan = dict(M=np.full(T, 8, np.int32), T=T)
    g._epoch_plan = lambda: plan
    g.loss_history, g.step_costs = [], []
    assert g.run_epoch(0) is not None
    m = g._model
    assert sum(n for _, n in m.calls) == T and all(n <= steps_per_call for _, n in m.calls)
    # (dev: the library reconciles small tables itself inside g4r_train_steps -- the host then neither cuts its calls nor syncs)
    every = 4 if sync_every == 'auto' else sync_every      # two ranks: 'auto' is 4 steps
    want = [k for k in range(every, T, every)] if (every and not dev) else []
    assert m.syncs == want
    if every:
        assert m.dev_k == every
    assert m.maxes == len(m.calls) + 1            # one collective NaN check per call + one for the epoch loss


def test_sync_every_auto_follows_rank_count_catalogue_size_and_the_exact_mode():
    """'auto': 4 steps at two ranks, 16 from three on (round-3 study, 2.5 K items); 64 for catalogues whose item tables do not fit
    the on-stream dense reconciliation (round-4 study at 1 M items x 256: the frequency does not move Recall@20 there, the volume
    grows with the interval); the exact-replica mode reconciles nothing."""
    from gru4rec_amd.gru4rec import GRU4Rec
    g = GRU4Rec(layers=[100])
    g.n_items = 37483
    assert (g.sync_steps(2), g.sync_steps(4), g.sync_steps(8)) == (4, 16, 16)
    g.n_items = 10_000_000
    g.layers = [256]
    assert (g.sync_steps(2), g.sync_steps(8)) == (64, 64)
    g.sync_every = 8
    assert g.sync_steps(8) == 8
    g.sync_every = 0
    assert g.sync_steps(8) == 0
    g.sync_every = 'auto'
    g.sparse_exact = True
    assert g.sync_steps(8) == 0
