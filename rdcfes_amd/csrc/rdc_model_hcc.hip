// rdc_model_hcc.hip — kernel instantiations of the Hcc model (see rdc_integrands.h for the citations)
#include "rdc_launch.h"
namespace rdc {
template hipError_t launch_rd<Hcc>(const LaunchArgs&, const Hcc::K&);
template hipError_t launch_rd<HccMassOnly>(const LaunchArgs&, const HccMassOnly::K&);
}
