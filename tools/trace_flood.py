import os, sys
os.environ["IVX_FLOOD_TRACE"]="1"
sys.path.insert(0,".")
import numpy as np
from bench import synth_v512, BONE
from invesalius3_amd.device import DeviceVolume
from scipy.ndimage import generate_binary_structure
img=synth_v512((512,512,512))
z,y,x=np.unravel_index(int(np.argmax(img)),img.shape)
vol=DeviceVolume(img)
vol.threshold(*BONE); vol.zero_out_mask()
r=vol.region_grow([(int(x),int(y),int(z))],BONE[0],BONE[1],generate_binary_structure(3,3),fill=1,select_value=254)
vol.sync(); print("rounds",r)
