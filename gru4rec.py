"""`import gru4rec; gru4rec.GRU4Rec` -- the module name the reference's users, `run.py -g gru4rec` and checkpoints
(pickles name the class `gru4rec.GRU4Rec`) expect.  The implementation is gru4rec_amd/gru4rec.py (MI355X / HIP)."""
from gru4rec_amd.gru4rec import GRU4Rec  # noqa: F401
