"""ctypes binding of libgru4rec_hip.so (C ABI: include/gru4rec_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m gru4rec_amd.build`.  There is no
CPU fallback: if the shared object is missing, or no MI355X is visible, the product path raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('G4R_LIB') or os.path.join(_HERE, 'libgru4rec_hip.so')   # G4R_LIB: developer override

G4R_MAX_LAYERS = 8
LOSS_IDS = {'cross-entropy': 0, 'bpr-max': 1, 'top1-max': 2, 'bpr': 3, 'top1': 4, 'xe_logit': 5}
ACT_IDS = {'linear': 0, 'relu': 1, 'tanh': 2, 'leaky': 3, 'elu': 4, 'selu': 5, 'softmax': 6, 'softmax_logit': 7}
ADAPT_IDS = {'adagrad': 0, 'rmsprop': 1, 'adadelta': 2, 'adam': 3, None: 4}
RANK_MODES = {'standard': 0, 'conservative': 1, 'median': 2, 'tiebreaking': 3}
EMBED_CONSTRAINED, EMBED_SEPARATE, EMBED_ONEHOT = 0, 1, 2


class G4RConfig(C.Structure):
    _fields_ = [
        ('n_items', C.c_int32), ('n_layers', C.c_int32), ('layers', C.c_int32 * G4R_MAX_LAYERS),
        ('batch_size', C.c_int32), ('n_sample', C.c_int32), ('loss', C.c_int32),
        ('final_act', C.c_int32), ('final_act_p0', C.c_float), ('final_act_p1', C.c_float),
        ('hidden_act', C.c_int32), ('hidden_act_p0', C.c_float), ('hidden_act_p1', C.c_float),
        ('embed_mode', C.c_int32), ('embedding', C.c_int32),
        ('learning_rate', C.c_float), ('momentum', C.c_float), ('lmbd', C.c_float), ('bpreg', C.c_float),
        ('logq', C.c_float), ('sample_alpha', C.c_float),
        ('dropout_p_hidden', C.c_float), ('dropout_p_embed', C.c_float),
        ('sample_store', C.c_int64), ('seed', C.c_uint64),
        ('device', C.c_int32), ('rank', C.c_int32), ('nranks', C.c_int32), ('use_graph', C.c_int32),
        ('smoothing', C.c_float), ('adapt', C.c_int32), ('adapt_p0', C.c_float), ('adapt_p1', C.c_float),
        ('grad_cap', C.c_float), ('sparse_exact', C.c_int32), ('defer_updates', C.c_int32),
    ]


# every symbol include/gru4rec_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'g4r_device_count', 'g4r_last_error', 'g4r_version', 'g4r_sizeof_config', 'g4r_create', 'g4r_destroy', 'g4r_set_param',
    'g4r_get_param', 'g4r_set_popularity', 'g4r_set_sample_store', 'g4r_get_sample_store',
    'g4r_sample_store_rows', 'g4r_build_plan', 'g4r_set_plan', 'g4r_train_steps', 'g4r_get_losses',
    'g4r_synchronize', 'g4r_global_step', 'g4r_refills', 'g4r_set_step_counters', 'g4r_kernel_time', 'g4r_profile', 'g4r_reset_hidden',
    'g4r_predict_begin', 'g4r_predict_hidden', 'g4r_predict_step', 'g4r_rank_targets', 'g4r_evaluate', 'g4r_comm_unique_id',
    'g4r_comm_init', 'g4r_virtual_train_steps', 'g4r_virtual_sync_dense', 'g4r_comm_sync_sparse', 'g4r_sync_set_rule', 'g4r_set_sync_every', 'g4r_comm_min_i64', 'g4r_comm_max_i64', 'g4r_comm_nranks', 'g4r_p2p_enable', 'g4r_p2p_export', 'g4r_p2p_attach', 'g4r_p2p_active', 'g4r_sync_enable', 'g4r_sync_row_floats', 'g4r_sync_export', 'g4r_sync_import', 'g4r_get_debug', 'g4r_stress_start', 'g4r_stress_stop', 'g4r_selftest_mfma', 'g4r_bench_rows',
    'g4r_events_load', 'g4r_events_rows', 'g4r_events_items', 'g4r_events_item_bytes', 'g4r_events_time_kind',
    'g4r_events_copy', 'g4r_events_free',
]

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises NativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError('libgru4rec_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
                          'g.build()"` -- the MI355X path has no CPU fallback.' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    L.g4r_device_count.restype = C.c_int
    L.g4r_last_error.restype = C.c_char_p
    L.g4r_version.restype = C.c_char_p
    L.g4r_create.argtypes = [C.POINTER(G4RConfig), C.POINTER(vp)]
    L.g4r_destroy.argtypes = [vp]
    L.g4r_destroy.restype = None
    L.g4r_set_param.argtypes = [vp, C.c_char_p, i32, f32p, i64]
    L.g4r_get_param.argtypes = [vp, C.c_char_p, i32, f32p, i64]
    L.g4r_set_popularity.argtypes = [vp, f32p, f32p, f32p, i64]
    L.g4r_set_sample_store.argtypes = [vp, i32p, i64]
    L.g4r_get_sample_store.argtypes = [vp, i32p, i64]
    L.g4r_sample_store_rows.argtypes = [vp]
    L.g4r_sample_store_rows.restype = i64
    L.g4r_build_plan.argtypes = [i32p, i64, i64p, i32p, i32, i32, i32p, i32p, u8p, i32p, i64p, i32p, i64, i64, i64p]
    L.g4r_build_plan.restype = i64
    L.g4r_set_plan.argtypes = [vp, i32p, i32p, u8p, i32p, i64, i64p, i32p, i64]
    L.g4r_train_steps.argtypes = [vp, i64, i64]
    L.g4r_get_losses.argtypes = [vp, i64, i64, f32p]
    L.g4r_synchronize.argtypes = [vp]
    L.g4r_global_step.argtypes = [vp]
    L.g4r_global_step.restype = i64
    L.g4r_refills.argtypes, L.g4r_refills.restype = [vp], i64
    L.g4r_set_step_counters.argtypes = [vp, i64, i64]
    L.g4r_kernel_time.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), i64p]
    L.g4r_profile.argtypes = [vp, i32]
    L.g4r_reset_hidden.argtypes = [vp]
    L.g4r_predict_begin.argtypes = [vp, i32]
    L.g4r_predict_hidden.argtypes = [vp, u8p, i32, i32p, i32]
    L.g4r_predict_step.argtypes = [vp, i32p, i32, i32p, i64, f32p]
    L.g4r_rank_targets.argtypes = [vp, i32p, i32, i64, i32, f32p]
    L.g4r_evaluate.argtypes = [vp, i32p, i32p, u8p, i32p, i64, i32, i64p, i32p, i64, i32p, i64, i32p, i32, i32,
                               C.POINTER(C.c_double), C.POINTER(C.c_double), i64p]
    L.g4r_comm_unique_id.argtypes = [C.c_char_p]
    L.g4r_comm_init.argtypes = [vp, C.c_char_p, i32, i32]
    L.g4r_comm_sync_sparse.argtypes = [vp]
    L.g4r_virtual_train_steps.argtypes = [C.POINTER(vp), i32, i64, i64]
    L.g4r_virtual_sync_dense.argtypes = [C.POINTER(vp), i32]
    L.g4r_sync_set_rule.argtypes = [vp, i32, i32]
    L.g4r_set_sync_every.argtypes = [vp, i32]
    L.g4r_comm_min_i64.argtypes = [vp, i64p]
    L.g4r_comm_max_i64.argtypes = [vp, i64p]
    L.g4r_comm_nranks.argtypes = [vp]
    L.g4r_p2p_enable.argtypes = [vp]
    L.g4r_p2p_export.argtypes = [vp, C.c_char_p]
    L.g4r_p2p_attach.argtypes = [vp, C.c_char_p, i32, i32]
    L.g4r_p2p_active.argtypes = [vp]
    L.g4r_sync_enable.argtypes = [vp]
    L.g4r_sync_row_floats.argtypes, L.g4r_sync_row_floats.restype = [vp, i32], i64
    L.g4r_sync_export.argtypes, L.g4r_sync_export.restype = [vp, i32, i32p, f32p, i64], i64
    L.g4r_sync_import.argtypes = [vp, i32, i32, i64p, C.POINTER(i32p), C.POINTER(f32p)]
    L.g4r_get_debug.argtypes = [vp, C.c_char_p, f32p, i64]
    L.g4r_selftest_mfma.argtypes = [f32p]
    L.g4r_stress_start.argtypes = [i32, i64, i32, C.POINTER(vp)]
    L.g4r_stress_stop.argtypes = [vp]
    L.g4r_bench_rows.argtypes = [i32, i64, i32, i64, i32, i32, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.g4r_events_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, i32, C.POINTER(vp)]
    for fn in (L.g4r_events_rows, L.g4r_events_items, L.g4r_events_item_bytes):
        fn.argtypes, fn.restype = [vp], i64
    L.g4r_events_time_kind.argtypes, L.g4r_events_time_kind.restype = [vp], i32
    L.g4r_events_copy.argtypes = [vp, i32p, i32p, vp, i64p, C.c_char_p]
    L.g4r_events_free.argtypes, L.g4r_events_free.restype = [vp], None
    if L.g4r_sizeof_config() != C.sizeof(G4RConfig):
        raise NativeError('g4r_config layout mismatch between the header and the ctypes binding')
    _lib = L
    return L


def device_count():
    return int(lib().g4r_device_count())


def _chk(rc):
    if rc != 0:
        raise NativeError(lib().g4r_last_error().decode())


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


IO_UNSUPPORTED = 1      # G4R_IO_UNSUPPORTED


def load_events(path, session_col, item_col, time_col=None, threads=0):
    """Native TSV parse (g4r_events_load): dict(session int32[n], item_idx int32[n], time int64|float64[n] or None,
    item_ids list[str] in index order = order of first appearance), or None when the file needs pandas' general parser."""
    L = lib()
    h = C.c_void_p()
    rc = L.g4r_events_load(os.fsencode(path), session_col.encode(), item_col.encode(),
                           None if time_col is None else time_col.encode(), threads, C.byref(h))
    if rc == IO_UNSUPPORTED:
        return None
    _chk(rc)
    try:
        n, k, nb, kind = L.g4r_events_rows(h), L.g4r_events_items(h), L.g4r_events_item_bytes(h), L.g4r_events_time_kind(h)
        session = np.empty(n, dtype=np.int32)
        item_idx = np.empty(n, dtype=np.int32)
        time = None if kind == 0 else np.empty(n, dtype=np.int64 if kind == 1 else np.float64)
        off = np.empty(k + 1, dtype=np.int64)
        raw = C.create_string_buffer(max(int(nb), 1))
        _chk(L.g4r_events_copy(h, _i32(session), _i32(item_idx), None if time is None else time.ctypes.data_as(C.c_void_p),
                               _i64(off), raw))
    finally:
        L.g4r_events_free(h)
    try:
        blob = raw.raw[:nb].decode('utf-8')
    except UnicodeDecodeError:
        return None      # item ids in another encoding: pandas' reader decides how to read them
    lo = off.tolist()
    if len(blob) == nb:      # pure ASCII: byte offsets are character offsets
        ids = [blob[lo[i]:lo[i + 1]] for i in range(k)]
    else:
        rb = raw.raw
        ids = [rb[lo[i]:lo[i + 1]].decode('utf-8') for i in range(k)]
    return dict(session=session, item_idx=item_idx, time=time, item_ids=ids)


def build_plan(offset_sessions, session_order, data_items, batch_size, n_sample):
    """Host scheduler (no GPU): the (X, Y, M, R) stream of one epoch, gru4rec.py:594-651."""
    L = lib()
    off = np.ascontiguousarray(offset_sessions, dtype=np.int32)
    order = np.ascontiguousarray(session_order, dtype=np.int64)
    items = np.ascontiguousarray(data_items, dtype=np.int32)
    n_sess = len(off) - 1
    nc = C.c_int64(0)
    T = L.g4r_build_plan(_i32(off), n_sess, _i64(order), _i32(items), batch_size, n_sample,
                         None, None, None, None, None, None, 0, 0, C.byref(nc))
    if T < 0:
        raise NativeError(L.g4r_last_error().decode())
    B = batch_size
    plan = dict(in_idx=np.zeros((max(T, 1), B), dtype=np.int32), out_idx=np.zeros((max(T, 1), B), dtype=np.int32),
                reset=np.zeros((max(T, 1), B), dtype=np.uint8), M=np.zeros(max(T, 1), dtype=np.int32),
                compact_steps=np.zeros(max(nc.value, 1), dtype=np.int64),
                compact_maps=np.full((max(nc.value, 1), B), -1, dtype=np.int32))
    nc2 = C.c_int64(0)
    T2 = L.g4r_build_plan(_i32(off), n_sess, _i64(order), _i32(items), batch_size, n_sample,
                          _i32(plan['in_idx']), _i32(plan['out_idx']), _u8(plan['reset']), _i32(plan['M']),
                          _i64(plan['compact_steps']), _i32(plan['compact_maps']), max(T, 1), max(nc.value, 1),
                          C.byref(nc2))
    if T2 != T:
        raise NativeError('plan builder is not deterministic: %s' % L.g4r_last_error().decode())
    plan['T'] = int(T)
    plan['n_compact'] = int(nc.value)
    for k in ('in_idx', 'out_idx', 'reset', 'M'):
        plan[k] = plan[k][:T]
    plan['compact_steps'] = plan['compact_steps'][:nc.value]
    plan['compact_maps'] = plan['compact_maps'][:nc.value]
    return plan


class Model:
    """Thin RAII wrapper over a g4r_model handle."""

    def __init__(self, **kw):
        L = lib()
        if L.g4r_device_count() <= 0:
            raise NativeError('no MI355X / HIP device visible: the gfx950 path has no CPU fallback')
        cfg = G4RConfig()
        layers = list(kw.pop('layers'))
        cfg.n_layers = len(layers)
        for i, d in enumerate(layers):
            cfg.layers[i] = int(d)
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.layers = layers
        self.h = C.c_void_p()
        _chk(L.g4r_create(C.byref(cfg), C.byref(self.h)))
        self.T = 0

    def close(self):
        if getattr(self, 'h', None):
            lib().g4r_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters
    def set_param(self, name, arr, layer=0):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        _chk(lib().g4r_set_param(self.h, name.encode(), layer, _f32(a), a.size))

    def get_param(self, name, shape, layer=0):
        a = np.empty(shape, dtype=np.float32)
        _chk(lib().g4r_get_param(self.h, name.encode(), layer, _f32(a), a.size))
        return a

    def get_debug(self, name, shape):
        a = np.empty(shape, dtype=np.float32)
        _chk(lib().g4r_get_debug(self.h, name.encode(), _f32(a), a.size))
        return a

    # -- sampling
    def set_popularity(self, cum_p, lq_tgt=None, lq_smp=None):
        p = np.ascontiguousarray(cum_p, dtype=np.float32)
        a = None if lq_tgt is None else np.ascontiguousarray(lq_tgt, dtype=np.float32)
        b = None if lq_smp is None else np.ascontiguousarray(lq_smp, dtype=np.float32)
        _chk(lib().g4r_set_popularity(self.h, _f32(p), None if a is None else _f32(a),
                                      None if b is None else _f32(b), p.size))

    def sample_store_rows(self):
        return int(lib().g4r_sample_store_rows(self.h))

    def set_sample_store(self, st):
        a = np.ascontiguousarray(st, dtype=np.int32)
        _chk(lib().g4r_set_sample_store(self.h, _i32(a), a.shape[0]))

    def get_sample_store(self, n_sample):
        rows = self.sample_store_rows()
        a = np.empty((rows, n_sample), dtype=np.int32)
        _chk(lib().g4r_get_sample_store(self.h, _i32(a), rows))
        return a

    # -- plan + training
    def set_plan(self, plan):
        nc = int(plan.get('n_compact', 0))
        cs = np.ascontiguousarray(plan['compact_steps'], dtype=np.int64) if nc else None
        cm = np.ascontiguousarray(plan['compact_maps'], dtype=np.int32) if nc else None
        ii = np.ascontiguousarray(plan['in_idx'], dtype=np.int32)
        oi = np.ascontiguousarray(plan['out_idx'], dtype=np.int32)
        rs = np.ascontiguousarray(plan['reset'], dtype=np.uint8)
        mm = np.ascontiguousarray(plan['M'], dtype=np.int32)
        _chk(lib().g4r_set_plan(self.h, _i32(ii), _i32(oi), _u8(rs), _i32(mm), len(mm),
                                None if cs is None else _i64(cs), None if cm is None else _i32(cm), nc))
        self.T = len(mm)

    def train_steps(self, t0, n):
        _chk(lib().g4r_train_steps(self.h, t0, n))

    def get_losses(self, t0, n):
        a = np.empty(n, dtype=np.float32)
        _chk(lib().g4r_get_losses(self.h, t0, n, _f32(a)))
        return a

    def reset_hidden(self):
        _chk(lib().g4r_reset_hidden(self.h))

    def global_step(self):
        return int(lib().g4r_global_step(self.h))

    def refills(self):
        return int(lib().g4r_refills(self.h))

    def set_step_counters(self, global_step, refills):
        _chk(lib().g4r_set_step_counters(self.h, int(global_step), int(refills)))

    def profile(self, enable):
        """False / True, or 2: profile with the update launch split into its two roles (k_dense_grad, k_sparse_update)."""
        _chk(lib().g4r_profile(self.h, 2 if enable == 2 else (1 if enable else 0)))

    def kernel_times(self):
        out = {}
        i = 0
        while True:
            name = C.c_char_p()
            ms = C.c_double()
            n = C.c_int64()
            if lib().g4r_kernel_time(self.h, i, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                break
            if n.value:
                out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    # -- prediction
    def predict_begin(self, batch):
        _chk(lib().g4r_predict_begin(self.h, batch))

    def predict_hidden(self, zero_mask=None, keep_rows=None):
        z = None if zero_mask is None else np.ascontiguousarray(zero_mask, dtype=np.uint8)
        k = None if keep_rows is None else np.ascontiguousarray(keep_rows, dtype=np.int32)
        _chk(lib().g4r_predict_hidden(self.h, None if z is None else _u8(z), 0 if z is None else len(z), None if k is None else _i32(k),
                                      0 if k is None else len(k)))

    def predict_step(self, in_idx, item_idx=None, want_scores=True):
        ii = np.ascontiguousarray(in_idx, dtype=np.int32)
        it = None if item_idx is None else np.ascontiguousarray(item_idx, dtype=np.int32)
        n_sel = self.cfg.n_items if it is None else len(it)
        out = np.empty((len(ii), n_sel), dtype=np.float32) if want_scores else None
        _chk(lib().g4r_predict_step(self.h, _i32(ii), len(ii), None if it is None else _i32(it), n_sel,
                                    None if out is None else _f32(out)))
        return out

    def rank_targets(self, target_col, col_begin=0, mode='standard'):
        t = np.ascontiguousarray(target_col, dtype=np.int32)
        r = np.empty(len(t), dtype=np.float32)
        _chk(lib().g4r_rank_targets(self.h, _i32(t), len(t), col_begin, RANK_MODES[mode], _f32(r)))
        return r

    # -- multi-GPU
    def evaluate(self, plan, batch, items, cutoffs, mode):
        """Whole evaluation in one call (g4r_evaluate).  Returns (recall_sum[n_cut], mrr_sum[n_cut], n_events)."""
        T = int(plan['T'])
        cuts = np.ascontiguousarray(cutoffs, dtype=np.int32)
        rec = np.zeros(len(cuts), dtype=np.float64)
        mrr = np.zeros(len(cuts), dtype=np.float64)
        n = C.c_int64(0)
        it = None if items is None else np.ascontiguousarray(items, dtype=np.int32)
        nc = int(plan.get('n_compact', 0))
        cs = np.ascontiguousarray(plan['compact_steps'], dtype=np.int64) if nc else np.zeros(1, dtype=np.int64)
        cm = np.ascontiguousarray(plan['compact_maps'], dtype=np.int32) if nc else np.zeros((1, batch), dtype=np.int32)
        arr = {k: np.ascontiguousarray(plan[k]) for k in ('in_idx', 'out_idx', 'reset', 'M')}
        _chk(lib().g4r_evaluate(self.h, _i32(arr['in_idx']), _i32(arr['out_idx']), _u8(arr['reset']), _i32(arr['M']), T, batch,
                                _i64(cs), _i32(cm), nc, None if it is None else _i32(it), 0 if it is None else len(it),
                                _i32(cuts), len(cuts), RANK_MODES[mode], rec.ctypes.data_as(C.POINTER(C.c_double)),
                                mrr.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)))
        return rec, mrr, int(n.value)

    def comm_init(self, unique_id, nranks, rank):
        _chk(lib().g4r_comm_init(self.h, unique_id, nranks, rank))

    def comm_sync_sparse(self):
        _chk(lib().g4r_comm_sync_sparse(self.h))

    def p2p_enable(self):
        """The step's dense-gradient all-reduce through peer memory instead of RCCL (g4r_p2p_enable; collective)."""
        _chk(lib().g4r_p2p_enable(self.h))

    def p2p_export(self):
        """This rank's 64-byte IPC handle of its exchange region (g4r_p2p_export)."""
        buf = C.create_string_buffer(64)
        _chk(lib().g4r_p2p_export(self.h, buf))
        return buf.raw

    def p2p_attach(self, handles, nranks, rank):
        """handles: the ranks' 64-byte handles in rank order (g4r_p2p_attach)."""
        blob = b''.join(handles)
        if len(blob) != 64 * nranks:
            raise ValueError('p2p_attach: %d handles of 64 bytes expected' % nranks)
        _chk(lib().g4r_p2p_attach(self.h, blob, nranks, rank))

    def p2p_active(self):
        return bool(lib().g4r_p2p_active(self.h))

    def sync_enable(self):
        _chk(lib().g4r_sync_enable(self.h))

    def set_sync_every(self, k):
        """True when g4r_train_steps reconciles the item tables itself every k steps (small tables, g4r_set_sync_every); False when
        the caller has to call comm_sync_sparse."""
        rc = lib().g4r_set_sync_every(self.h, int(k))
        if rc < 0:
            raise NativeError(lib().g4r_last_error().decode())
        return rc == 1

    def sync_set_rule(self, param_rule, stat_rule):
        """'sum' / 'mean' for the parameter planes and for the optimizer-statistic planes of the reconciliation (g4r_sync_set_rule)."""
        r = {'sum': 0, 'mean': 1}
        _chk(lib().g4r_sync_set_rule(self.h, r[param_rule], r[stat_rule]))

    def sync_export(self, group=0):
        """(sorted ids int32[n], delta rows float32[n * row_floats] plane after plane) of the rows touched since the last sync."""
        n = int(lib().g4r_sync_export(self.h, group, None, None, 0))
        if n < 0:
            raise NativeError(lib().g4r_last_error().decode())
        w = int(lib().g4r_sync_row_floats(self.h, group))
        ids = np.empty(n, dtype=np.int32)
        rows = np.empty(n * w, dtype=np.float32)
        if lib().g4r_sync_export(self.h, group, _i32(ids), _f32(rows), n) != n:
            raise NativeError(lib().g4r_last_error().decode())
        return ids, rows

    def sync_import(self, parts, group=0):
        """parts: [(ids, rows)] of ALL ranks in rank order."""
        n = len(parts)
        counts = np.array([len(p[0]) for p in parts], dtype=np.int64)
        ids = [np.ascontiguousarray(p[0], dtype=np.int32) for p in parts]
        rows = [np.ascontiguousarray(p[1], dtype=np.float32) for p in parts]
        pi = (C.POINTER(C.c_int32) * n)(*[_i32(a) for a in ids])
        pr = (C.POINTER(C.c_float) * n)(*[_f32(a) for a in rows])
        _chk(lib().g4r_sync_import(self.h, group, n, _i64(counts), pi, pr))

    def comm_min(self, value):
        v = C.c_int64(int(value))
        _chk(lib().g4r_comm_min_i64(self.h, C.byref(v)))
        return int(v.value)

    def comm_max(self, value):
        v = C.c_int64(int(value))
        _chk(lib().g4r_comm_max_i64(self.h, C.byref(v)))
        return int(v.value)

    def comm_nranks(self):
        n = int(lib().g4r_comm_nranks(self.h))
        if n < 0:
            raise NativeError(lib().g4r_last_error().decode())
        return n


def virtual_train_steps(models, t0, n):
    """Plan steps [t0, t0 + n) of `models` (handle q = rank q of len(models) ranks on one device) in lock-step, the dense gradients
    summed in process where the real run all-reduces them (g4r_virtual_train_steps)."""
    hs = (C.c_void_p * len(models))(*[m.h for m in models])
    _chk(lib().g4r_virtual_train_steps(hs, len(models), int(t0), int(n)))


def virtual_sync_dense(models):
    """Dense reconciliation of the item tables of `models` (handle q = rank q), the all-reduce taken in process (g4r_virtual_sync_dense)."""
    hs = (C.c_void_p * len(models))(*[m.h for m in models])
    _chk(lib().g4r_virtual_sync_dense(hs, len(models)))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    _chk(lib().g4r_comm_unique_id(buf))
    return buf.raw


def bench_rows(n_items, width, rows_per_launch, launches=200, mode=1, device=0, seed=1):
    """(mean kernel us, wall us per launch) of the row gather / scatter micro-benchmark (g4r_bench_rows)."""
    k, w = C.c_double(), C.c_double()
    _chk(lib().g4r_bench_rows(device, int(n_items), int(width), int(rows_per_launch), int(launches), int(mode), int(seed), C.byref(k), C.byref(w)))
    return k.value, w.value


def selftest_mfma():
    e = C.c_float()
    _chk(lib().g4r_selftest_mfma(C.byref(e)))
    return e.value


class MemoryStress:
    """HBM / Infinity-Cache load on a stream of its own while the `with` body runs (g4r_stress_start / g4r_stress_stop)."""

    def __init__(self, mbytes=4096, launches=400, device=0):
        self.args = (int(device), int(mbytes), int(launches))
        self.h = None

    def __enter__(self):
        h = C.c_void_p()
        _chk(lib().g4r_stress_start(self.args[0], self.args[1], self.args[2], C.byref(h)))
        self.h = h
        return self

    def __exit__(self, *exc):
        if self.h is not None:
            _chk(lib().g4r_stress_stop(self.h))
            self.h = None
        return False
