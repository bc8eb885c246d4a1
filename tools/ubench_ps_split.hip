// tools/ubench_ps.hip with the operand products of a general column as a chain of their own (PGPU_PS_SPLIT=1, csrc/hensel_ps.hpp)
#define PGPU_PS_SPLIT 1
#include "ubench_ps.hip"
