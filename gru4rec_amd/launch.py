"""One process per GPU, without torch: rank layout, RCCL rendezvous and the tiny control plane of a data-parallel run.

The reference is single-process / single-GPU (run.py:101-113 trains on `device=cuda0`); north_star adds "sessions shard naturally
across the 8 GPUs of one node".  Ranks are ordinary processes that find RANK / LOCAL_RANK / WORLD_SIZE in their environment -- set
by `spawn()` below (`bench.py --gpus N`, `run.py --gpus N`) or by `python -m torch.distributed.run` (the driver's launcher; nothing of
torch is imported here).  The only thing ranks have to agree on before RCCL exists is the 128-byte unique id: rank 0 creates it
(`ncclGetUniqueId`, whose bootstrap root thread then lives in rank 0) and publishes it through a file in the temp directory named
after the launcher's pid and rendezvous token; the other ranks poll for it.  Everything after that -- barriers, the max over ranks
of the timed region, the collective NaN exit, the common plan length -- is `g4r_comm_max_i64` on the communicator itself."""
import os
import subprocess
import sys
import tempfile
import time


def layout():
    """(rank, world, local_rank) from the environment (1 process = (0, 1, 0))."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))


def _rendezvous_file():
    token = os.environ.get('G4R_RDZV') or '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
    return os.path.join(tempfile.gettempdir(), 'g4r_uid_%d_%s' % (os.getppid(), token))


def unique_id(rank, world, timeout=300.0, make=None):
    """The RCCL unique id of this run: created by rank 0, read by the others (file rendezvous keyed by the common parent process).
    make: what rank 0 calls to create the 128 bytes (default g4r_comm_unique_id = ncclGetUniqueId; tests without a GPU pass their own)."""
    if make is None:
        from . import _native
        make = _native.comm_unique_id
    if world <= 1:
        # G4R_FORCE_STAGED=1: the N > 1 data path with a one-rank communicator (what a 1-GPU box can run of it)
        return make() if os.environ.get('G4R_FORCE_STAGED') else None
    path = _rendezvous_file()
    if rank == 0:
        uid = make()
        tmp = '%s.%d' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)      # atomic: a reader sees all 128 bytes or no file
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                uid = f.read()
            if len(uid) == 128:
                return uid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError('rank %d: no RCCL unique id from rank 0 after %.0f s (%s)' % (rank, timeout, path))
        time.sleep(0.02)


def cleanup(rank):
    """Rank 0, once every rank holds a communicator (i.e. behind a barrier): remove the rendezvous file."""
    if rank == 0:
        try:
            os.remove(_rendezvous_file())
        except OSError:
            pass


def barrier(model):
    model.comm_max(0)


def max_over_ranks_us(model, seconds):
    """max over ranks of a duration, through the int64 max-reduce (microsecond resolution)."""
    return model.comm_max(int(round(seconds * 1e6))) / 1e6


def gather_us(model, rank, world, seconds):
    """[duration of every rank]: world max-reduces, rank r contributing its value to the r-th and 0 to the others."""
    return [model.comm_max(int(round(seconds * 1e6)) if r == rank else 0) / 1e6 for r in range(world)]


def spawn(script, argv, n, quiet_ranks=True):
    """Run `script argv` as n ranks on this node (RANK = LOCAL_RANK = 0..n-1).  Rank 0 keeps this process's stdout; the other ranks'
    stdout goes to stderr (quiet_ranks: a rank's JSON / progress lines must not mix with rank 0's, its errors must stay visible).
    When a rank exits non-zero the others are terminated (they would block in RCCL).  Returns the exit code."""
    token = '%d_%d' % (os.getpid(), int(time.time() * 1e6) & 0xFFFFFFF)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), G4R_RDZV=token)
        env.setdefault('MASTER_ADDR', '127.0.0.1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if (r == 0 or not quiet_ranks) else sys.stderr))
    rc = 0
    alive = set(range(n))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    sys.stderr.write('%s: rank %d exited with code %d\n' % (os.path.basename(script), r, code))
                    rc = rc or code or 1
                    for q in alive:      # the survivors wait for the dead rank inside RCCL: stop them
                        procs[q].terminate()
            if alive:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.remove(os.path.join(tempfile.gettempdir(), 'g4r_uid_%d_%s' % (os.getpid(), token)))
        except OSError:
            pass
    return rc
