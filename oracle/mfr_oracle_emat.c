/* placeholder; E-matrix oracle follows */
#include "mfr_oracle.h"
