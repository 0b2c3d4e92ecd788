"""BASELINE config 4 on one GPU: its calls on one stream and dealt to two / three standing engines (HIP streams), over X and from the folds' Grams."""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/sparse-lm_amd']
import bench
from sparselm_amd import _engine
c4 = bench.Config4(_engine.get_engine(0), 100_000, 5_000)
calls = c4.calls_of(1, 0)
c4.run(calls)
print("over X, one stream   %.4f s" % min(c4.run(calls)[0] for _ in range(3)))
for k in (2, 3, 4):
    c4.drop_streams(); c4.add_streams(k); c4.run(calls, k)
    print("over X, %d streams    %.4f s" % (k, min(c4.run(calls, k)[0] for _ in range(3))))
c4.drop_streams()
print("Grams built in %.3f s" % c4.build_covariance())
c4.run(calls)
print("Grams, one stream    %.4f s" % min(c4.run(calls)[0] for _ in range(3)))
for k in (2, 3, 4):
    c4.drop_streams(); c4.add_streams(k); c4.run(calls, k)
    print("Grams, %d streams     %.4f s" % (k, min(c4.run(calls, k)[0] for _ in range(3))))
c4.close()
