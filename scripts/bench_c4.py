#!/usr/bin/env python
"""The config-4 lines of scripts/bench_configs.py by themselves (log_prob_all, ImportanceFlowProposal.draw / update_log_q)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as B  # noqa: E402

B.run_c4()
B.run_c4_draw()
