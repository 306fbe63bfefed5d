"""The grid search as one launch (SURVEY 8(d) secondary run / row N2: evaluate_text_alignment.py:181-198): 2 187
page-sized problems, one scoring system each; a few timed steps for a kernel trace.  Usage: python tools/grid_search_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from text_alignment_amd import textSeqCompare as tsc

print(bench.nw_grid_search(tsc, torch))
