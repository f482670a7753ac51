#!/usr/bin/env python3
"""bench.py's `secondary.small_graph_epoch_ms` alone (captured / eager epoch time of main.run on the bundled graphs)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

print(json.dumps(bench.small_graph_epochs()))
