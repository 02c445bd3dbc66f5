// rdc_model_pihna.hip — kernel instantiations of the Pihna model (see rdc_integrands.h for the citations)
#include "rdc_launch.h"
namespace rdc {
template hipError_t launch_rd<Pihna>(const LaunchArgs&, const Pihna::K&);
}
