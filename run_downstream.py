#!/usr/bin/env python
"""Repo-root shim for the reference's entry point name: `python run_downstream.py --ds-train --simu-exp --ds-task TDOA ...`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sarssl_boot  # noqa: E402,F401
from sar_ssl_amd.run_downstream import main  # noqa: E402

if __name__ == "__main__":
    main()
