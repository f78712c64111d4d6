# python stamps_cfg.py <2|3> <cum|hits|count>   (diagnostic: swaps in the stamps build on the box copy)
import sys, os, shutil
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shutil.copy(ROOT+"/scratch/"+os.environ.get("STAMPS_LIB","lib_stamps.so"), ROOT+"/grace-devel_amd/lib/libgrace_hip.so")
sys.argv=[sys.argv[0]]+sys.argv[1:3]+["1"]
exec(open(os.path.join(ROOT,'profiles','recipes','prof_cfg.py')).read())
