"""Command-line front end of gru4rec_amd/isa_audit.py (the audit build.py runs on every library it links):
    python tools/isa_audit.py [file.s]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gru4rec_amd.isa_audit import *          # noqa: F401,F403,E402
from gru4rec_amd.isa_audit import main      # noqa: E402

if __name__ == '__main__':
    main()
