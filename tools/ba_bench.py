import json, sys
sys.path.insert(0, '/root/repo')
from orthosfm_amd import ba
print(json.dumps(ba.bench_global_ba()))
