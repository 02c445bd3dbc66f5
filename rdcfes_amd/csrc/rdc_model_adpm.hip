// rdc_model_adpm.hip — kernel instantiations of the Adpm model (see rdc_integrands.h for the citations)
#include "rdc_launch.h"
namespace rdc {
template hipError_t launch_rd<Adpm>(const LaunchArgs&, const Adpm::K&);
template hipError_t launch_rd<AdpmDecayOnly>(const LaunchArgs&, const AdpmDecayOnly::K&);
}
