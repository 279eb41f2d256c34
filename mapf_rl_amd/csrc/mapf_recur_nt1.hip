// mapf_recur_nt1.hip -- csrc/mapf_recur.hip built for ONE agent tile (environments / windows of up to 16 agents): the reference's own
// training shapes (<= 6 agents, config.max_num_agetns) and every curriculum level.  The kernels' GEMMs, cells and attention loop over
// agent tiles; with three tiles (48 agents) a 6-agent window pays three times the MFMA, LDS and pointwise work per step.
#define MAPF_RECUR_NT 1
#define MAPF_RECUR_SUFFIX _nt1
#include "mapf_recur.hip"
