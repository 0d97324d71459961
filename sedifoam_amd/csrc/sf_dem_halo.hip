// sf_dem_halo.hip -- ghost-particle halo of the 1-D slab decomposition and particle injection/removal.
#include <algorithm>
#include <vector>

#include "sf_dem.h"

namespace sf {

long long DemEngine::border_pack(int, double, double*, long long) { fail("border_pack: not implemented yet"); }
void DemEngine::border_unpack(int, const double*, long long) { fail("border_unpack: not implemented yet"); }
long long DemEngine::forward_pack(int, double, double*) { fail("forward_pack: not implemented yet"); }
void DemEngine::forward_unpack(int, const double*, long long) { fail("forward_unpack: not implemented yet"); }
long long DemEngine::migrate_pack(int, double, double*, long long) { fail("migrate_pack: not implemented yet"); }
void DemEngine::migrate_unpack(const double*, long long) { fail("migrate_unpack: not implemented yet"); }
int DemEngine::migrate_record_doubles() const { return 0; }
void DemEngine::create_particles(int, const double*, const double*, double, double, int, const double*)
{
  fail("lammps_create_particle: not implemented yet");
}
void DemEngine::delete_particles(const int*, int) { fail("lammps_delete_particle: not implemented yet"); }

}  // namespace sf
