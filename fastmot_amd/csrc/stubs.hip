// placeholders until net.hip / flow.hip land
#include "common.h"
