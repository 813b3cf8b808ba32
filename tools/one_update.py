import sys; sys.path.insert(0,'/root/repo')
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
prob = synth.make_problem(30, 2000, 10, seed=0)
eng = UpdateEngine(max_clones=30, max_features=2000, max_track=10)
eng.load(prob); eng.run(); eng.sync(); eng.run(); eng.sync()
