// rdc_model_proteas.hip — kernel instantiations of the Proteas model (see rdc_integrands.h for the citations)
#include "rdc_launch.h"
namespace rdc {
template hipError_t launch_rd<Proteas>(const LaunchArgs&, const Proteas::K&);
}
