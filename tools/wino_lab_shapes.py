import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
os.environ['TG_CONV_WINO'] = '1'
import wino_lab as W
for shp in [(38, 512, 512, 16, 16), (38, 256, 512, 16, 16), (38, 512, 512, 8, 8), (36, 64, 128, 16, 16), (36, 128, 128, 16, 16), (36, 32, 64, 16, 16), (12, 64, 64, 16, 16)]:
    W.bench(*shp)
