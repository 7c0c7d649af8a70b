// placeholders until net.hip / flow.hip land
#include "common.h"
void fm_flow_free(FlowState*) {}
