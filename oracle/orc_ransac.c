#include "mdrp_oracle.h"
