cd $GRAFT_REPO_ROOT
TACT=20 timeout -k 10 200 python tools/profile_actor.py 2>&1 | tail -2
TACT=3 timeout -k 10 200 python tools/profile_actor.py 2>&1 | tail -1
