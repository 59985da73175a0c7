import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import latency
latency.single('maze2', 1000, 8)
latency.single('kuka7', 2000, 10)
