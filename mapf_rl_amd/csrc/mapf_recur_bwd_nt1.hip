// mapf_recur_bwd_nt1.hip -- csrc/mapf_recur_bwd.hip built for ONE agent tile (see mapf_recur_nt1.hip).
#define MAPF_RECUR_NT 1
#define MAPF_RECUR_SUFFIX _nt1
#include "mapf_recur_bwd.hip"
