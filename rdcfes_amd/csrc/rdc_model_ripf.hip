// rdc_model_ripf.hip — kernel instantiations of the Ripf model (see rdc_integrands.h for the citations)
#include "rdc_launch.h"
namespace rdc {
template hipError_t launch_rd<Ripf>(const LaunchArgs&, const Ripf::K&);
}
