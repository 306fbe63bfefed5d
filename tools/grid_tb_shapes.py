"""Traceback launch shapes on the grid search's batch (2 187 problems of 800 x 900, a scoring system per problem):
fill / traceback time with the library's choice and with TA_NW_TBWAVES forced to 1, 3 (half-strip pairs), 2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_alignment_amd import textSeqCompare as tsc
from tools.nw_configs import grid_systems, time_batch
from tools.synth import synth_pair_ids
grid = grid_systems(); n, m = 800, 900
pages = [synth_pair_ids(n, m, 4400 + k) for k in range(3)]
params = np.array(grid * 3, dtype=np.int64)
nprob = len(params)
batch = tsc.NWBatch([pages[k // len(grid)][0] for k in range(nprob)], [pages[k // len(grid)][1] for k in range(nprob)], params)
for w in (None, 1, 3, 2):
    batch.tb_waves = w
    print(w, time_batch(torch, batch))
