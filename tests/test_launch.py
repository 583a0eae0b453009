"""The torch-free launcher of the N > 1 runs (gru4rec_amd/launch.py) as two real processes on the CPU: rank layout from the
environment, the file rendezvous of the 128-byte communicator id (rank 0 writes atomically, the others poll), rank 0 keeping
stdout, and the failure path -- a rank that dies takes the others down instead of leaving them waiting in a collective."""
import os
import subprocess
import sys
import textwrap
import time

from gru4rec_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / 'rank.py'
    p.write_text('import os, sys, time\nsys.path.insert(0, %r)\nfrom gru4rec_amd import launch\n' % ROOT + textwrap.dedent(body))
    return str(p)


def test_two_ranks_meet_through_the_file_rendezvous(tmp_path):
    s = _script(tmp_path, '''
        rank, world, local = launch.layout()
        if rank == 1:
            time.sleep(0.3)                      # the reader may well arrive first or last
        uid = launch.unique_id(rank, world, make=lambda: bytes(range(128)))
        open(os.path.join(sys.argv[1], 'rank%d' % rank), 'w').write('%d %d %d %s' % (rank, world, local, uid.hex()))
        print('hello from rank %d' % rank)
        launch.cleanup(rank) if rank == 0 and time.sleep(0.6) is None else None
    ''')
    r = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from gru4rec_amd import launch; '
                        'sys.exit(launch.spawn(%r, [%r], 2))' % (ROOT, s, str(tmp_path))], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    a = (tmp_path / 'rank0').read_text().split()
    b = (tmp_path / 'rank1').read_text().split()
    assert a[:3] == ['0', '2', '0'] and b[:3] == ['1', '2', '1']
    assert a[3] == b[3] == bytes(range(128)).hex()
    assert r.stdout.strip() == 'hello from rank 0'          # rank 0 owns stdout (one JSON line in bench.py) ...
    assert 'hello from rank 1' in r.stderr                  # ... the other ranks stay visible on stderr


def test_a_dead_rank_stops_the_others(tmp_path):
    s = _script(tmp_path, '''
        rank, world, _ = launch.layout()
        if rank == 1:
            sys.exit(3)
        time.sleep(120)                          # rank 0 "waits in a collective"
    ''')
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from gru4rec_amd import launch; '
                        'sys.exit(launch.spawn(%r, [], 2))' % (ROOT, s)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    assert 'rank 1 exited with code 3' in r.stderr
    assert time.time() - t0 < 30


def test_layout_defaults_to_one_process(monkeypatch):
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    assert launch.layout() == (0, 1, 0)
    assert launch.unique_id(0, 1, make=lambda: b'x' * 128) is None


def test_a_stale_or_foreign_rendezvous_file_is_not_accepted(tmp_path, monkeypatch):
    """Advisor finding of round 3: a file a crashed run left behind (or one planted by somebody else) must not be read as this run's
    id -- ncclCommInitRank would wait for a dead root.  The record carries a magic word and rank 0's clock; rank 0 removes an old
    file before it creates the id; the directory is private (0700)."""
    import struct
    d = tmp_path / 'rdzv'
    d.mkdir(mode=0o700)
    monkeypatch.setenv('G4R_RDZV_DIR', str(d))
    monkeypatch.setenv('G4R_RDZV', 'tok')
    path = launch._rendezvous_file()
    assert os.path.dirname(path) == str(d)
    # (a) a bare 128-byte file (the round-3 format), (b) a well-formed record stamped long before this process started
    for rec in (b'\x01' * 128, launch._MAGIC + struct.pack('<d', time.time() - 7200.0) + b'\x02' * 128):
        with open(path, 'wb') as f:
            f.write(rec)
        t0 = time.time()
        try:
            launch.unique_id(1, 2, timeout=0.3, make=lambda: b'')
            raise AssertionError('accepted a stale rendezvous file')
        except RuntimeError as e:
            assert 'no RCCL unique id' in str(e) and time.time() - t0 < 5
    # rank 0 replaces whatever is there; a reader then gets exactly its bytes
    uid = launch.unique_id(0, 2, make=lambda: bytes(range(128)))
    assert launch.unique_id(1, 2, timeout=5.0, make=lambda: b'') == uid == bytes(range(128))
    assert os.stat(path).st_mode & 0o077 == 0
    launch.cleanup(0)
    assert not os.path.exists(path)
    # without a launcher-made directory the fallback is a private per-user directory
    monkeypatch.delenv('G4R_RDZV_DIR')
    fb = launch._rendezvous_dir()
    assert os.stat(fb).st_mode & 0o077 == 0 and os.stat(fb).st_uid == os.getuid()
