#include "lammps_mock.h"
