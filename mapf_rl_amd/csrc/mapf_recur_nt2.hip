// mapf_recur_nt2.hip -- csrc/mapf_recur.hip built for TWO agent tiles (environments / windows of 17..32 agents), see mapf_recur_nt1.hip.
#define MAPF_RECUR_NT 2
#define MAPF_RECUR_SUFFIX _nt2
#include "mapf_recur.hip"
