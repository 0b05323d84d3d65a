import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load('/tmp/y_noks.npy')
print(sys.argv[1], "max rel vs no split-K: %.3e" % (np.abs(a - b).max() / np.abs(b).max()))
