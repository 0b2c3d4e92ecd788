import cProfile, pstats, io, runpy, sys, os
sys.argv = ["config4_gridsearch.py"]
# run once normally (rep 0 warms), then profile a third fit
ns = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "config4_gridsearch.py"))
import warnings
from sklearn.model_selection import KFold
from sparselm_amd.model import SparseGroupLasso
from sparselm_amd.model_selection import GridSearchCV
X, y, groups, grid = ns["X"], ns["y"], ns["groups"], ns["grid"]
pr = cProfile.Profile()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pr.enable()
    GridSearchCV(SparseGroupLasso(groups=groups), grid, cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
