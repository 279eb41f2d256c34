cd $GRAFT_REPO_ROOT
for shp in "4096 40 16" "8192 20 6" "4096 32 40" "65536 20 6"; do set -- $shp; echo "== $1 x ${2}x${2} / $3"; TE=$1 TL=$2 TN=$3 timeout -k 10 100 python tools/stamps.py 2>&1 | grep -v amdgpu; done
