# python with_lib.py <lib under scratch/> <script> [args]: runs script in a child with that library swapped in (box copy only)
import sys, os, shutil, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dst=ROOT+"/grace-devel_amd/lib/libgrace_hip.so"
shutil.copy(dst, ROOT+"/scratch/lib_cur.so")
shutil.copy(ROOT+"/scratch/"+sys.argv[1], dst)
rc=subprocess.call([sys.executable]+sys.argv[2:])
shutil.copy(ROOT+"/scratch/lib_cur.so", dst)
sys.exit(rc)
