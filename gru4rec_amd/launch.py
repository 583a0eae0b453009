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


_MAGIC = b'G4RUID01'
_T_START = time.time()      # when this rank's process imported the launcher: a rendezvous file must not be much older


def _rendezvous_dir():
    """A directory only this user can enter (0700): `spawn` makes a fresh one per launch (G4R_RDZV_DIR); under another launcher
    (torch.distributed.run) it is g4r_<uid> in the temp directory, created on first use and checked for owner and mode."""
    d = os.environ.get('G4R_RDZV_DIR')
    if d:
        return d
    d = os.path.join(tempfile.gettempdir(), 'g4r_%d' % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError('rendezvous directory %s is not a private directory of this user' % d)
    return d


def _rendezvous_file():
    token = os.environ.get('G4R_RDZV') or '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
    return os.path.join(_rendezvous_dir(), 'g4r_uid_%d_%s' % (os.getppid(), token))


def unique_id(rank, world, timeout=300.0, make=None, max_age=600.0):
    """The RCCL unique id of this run: created by rank 0, read by the others (file rendezvous keyed by the common parent process).
    make: what rank 0 calls to create the 128 bytes (default g4r_comm_unique_id = ncclGetUniqueId; tests without a GPU pass their own).
    The file lives in a 0700 directory and carries a magic word and rank 0's wall-clock time; rank 0 removes whatever a crashed
    earlier run left under the same name BEFORE it creates the id, and a reader ignores a file stamped more than `max_age`
    seconds before its own start (a stale id would leave ncclCommInitRank waiting for a dead root)."""
    if make is None:
        from . import _native
        make = _native.comm_unique_id
    if world <= 1:
        # G4R_FORCE_STAGED=1: the N > 1 data path with a one-rank communicator (what a 1-GPU box can run of it)
        return make() if os.environ.get('G4R_FORCE_STAGED') else None
    import struct
    path = _rendezvous_file()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
        uid = make()
        tmp = '%s.%d' % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(_MAGIC + struct.pack('<d', time.time()) + uid)
        os.replace(tmp, path)      # atomic: a reader sees the whole record or no file
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                rec = f.read()
            if len(rec) == 144 and rec[:8] == _MAGIC and struct.unpack('<d', rec[8:16])[0] >= _T_START - max_age:
                return rec[16:]
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
    rdir = tempfile.mkdtemp(prefix='g4r_rdzv_')      # 0700, fresh per launch: no stale id, nobody else can plant one
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), G4R_RDZV=token, G4R_RDZV_DIR=rdir)
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
        import shutil
        shutil.rmtree(rdir, ignore_errors=True)
    return rc
