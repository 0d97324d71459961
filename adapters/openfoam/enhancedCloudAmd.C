// enhancedCloudAmd.C -- enhancedCloud (enhancedCloud.H:183-249) forwarding to sf_cloud_* ; see enhancedCloudAmd.H.
#include "enhancedCloudAmd.H"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace Foam
{

// * * * * * * * * * * * * * * * * Local helpers * * * * * * * * * * * * * * //

static void sfCheck(int rc, const char* what)
{
    if (rc < 0)
    {
        FatalErrorIn(what) << sf_last_error() << abort(FatalError);
    }
}

//- Sorted distinct values of one coordinate of the cell centres (grid lines of a tensor-product mesh)
static void distinctSorted(const vectorField& c, const direction cmpt, const scalar tol, std::vector<scalar>& out)
{
    std::vector<scalar> v(c.size());
    forAll(c, i)
    {
        v[i] = c[i].component(cmpt);
    }
    std::sort(v.begin(), v.end());
    out.clear();
    for (std::size_t i = 0; i < v.size(); i++)
    {
        if (out.empty() || v[i] - out.back() > tol)
        {
            out.push_back(v[i]);
        }
    }
}

// * * * * * * * * * * * * * Private Member Functions  * * * * * * * * * * * //

// softParticleCloud::initLammps (softParticleCloud.C:43-117): the LAMMPS object on a duplicate of the world
// communicator, every line of in.lammps to every rank.  (The time-step adjustment the reference makes at the
// `timestep` line, adjustLampTimestep :209-261, is done by sf_cloud_create from the fluid time step.)
void enhancedCloud::initLammps()
{
    MPI_Comm commLammps;
    MPI_Comm_dup(MPI_COMM_WORLD, &commLammps);
    lmp_ = new LAMMPS_NS::LAMMPS(0, NULL, commLammps);

    FILE* fp = NULL;
    if (Pstream::master())
    {
        fp = std::fopen("in.lammps", "r");
        if (fp == NULL)
        {
            std::printf("initLammps::ERROR: Could not open LAMMPS input script.\n");
            MPI_Abort(MPI_COMM_WORLD, 1);
        }
    }
    Info<< "Reading Lammps inputfile (in.lammps) ..." << endl;
    lammps_sync(lmp_);

    int n = 0;
    char line[1024];
    while (1)
    {
        if (Pstream::master())
        {
            n = (std::fgets(line, 1024, fp) == NULL) ? 0 : int(std::strlen(line)) + 1;
            if (n == 0) std::fclose(fp);
        }
        Pstream::scatter(n);
        if (n == 0) break;
        MPI_Bcast(line, n, MPI_CHAR, 0, MPI_COMM_WORLD);
        lmp_->input->one(line);
    }
    Info<< "Finished reading Lammps inputfile." << endl;
    Info<< "FOAM reported # of particles according to Lammps: " << lammps_get_global_n(lmp_) << endl;
}


// The block's grid lines and OpenFOAM's numbering of its cells, from the cell centres of all processors.
// Global label of a cell = (cells of the lower-numbered processors) + its local label.
void enhancedCloud::describeMesh(sf_cloud_mesh& m)
{
    const label nprocs = Pstream::nProcs();
    const label me = Pstream::myProcNo();

    List<vectorField> centres(nprocs);
    centres[me] = mesh_.C().internalField();
    if (Pstream::parRun())
    {
        Pstream::gatherList(centres);
        Pstream::scatterList(centres);
    }
    procCells_.setSize(nprocs);
    nCellsGlobal_ = 0;
    forAll(centres, p)
    {
        if (p == me) cellOffset_ = nCellsGlobal_;
        procCells_[p] = centres[p].size();
        nCellsGlobal_ += centres[p].size();
    }
    vectorField all(nCellsGlobal_);
    {
        label k = 0;
        forAll(centres, p)
        {
            forAll(centres[p], i)
            {
                all[k++] = centres[p][i];
            }
        }
    }

    // bounding box of the whole mesh
    vector lo = mesh_.bounds().min();
    vector hi = mesh_.bounds().max();
    reduce(lo, minOp<vector>());
    reduce(hi, maxOp<vector>());

    std::vector<scalar> line[3];
    label n[3];
    for (direction d = 0; d < 3; d++)
    {
        const scalar tol = 1e-9*(hi.component(d) - lo.component(d));
        distinctSorted(all, d, tol, line[d]);
        n[d] = label(line[d].size());
        // faces from the centres: f[0] = low bound, f[i+1] = 2 c[i] - f[i]
        faces_[d].setSize(n[d] + 1);
        faces_[d][0] = lo.component(d);
        for (label i = 0; i < n[d]; i++)
        {
            faces_[d][i + 1] = 2.0*line[d][i] - faces_[d][i];
        }
        if (mag(faces_[d][n[d]] - hi.component(d)) > 1e-6*(hi.component(d) - lo.component(d)))
        {
            FatalErrorIn("enhancedCloud::describeMesh")
                << "the cell centres along axis " << label(d) << " do not tile the bounding box: "
                << "the mesh is not one rectilinear block" << abort(FatalError);
        }
        faces_[d][n[d]] = hi.component(d);
    }
    if (n[0]*n[1]*n[2] != nCellsGlobal_)
    {
        FatalErrorIn("enhancedCloud::describeMesh")
            << n[0] << " x " << n[1] << " x " << n[2] << " grid lines for " << nCellsGlobal_
            << " cells: the mesh is not one rectilinear block" << abort(FatalError);
    }

    cellLabel_.setSize(nCellsGlobal_);
    cellLabel_ = -1;
    forAll(all, g)
    {
        label idx[3];
        for (direction d = 0; d < 3; d++)
        {
            const scalar tol = 1e-9*(hi.component(d) - lo.component(d));
            idx[d] = label(std::lower_bound(line[d].begin(), line[d].end(), all[g].component(d) - tol) - line[d].begin());
        }
        cellLabel_[idx[0] + n[0]*(idx[1] + n[1]*idx[2])] = g;
    }
    forAll(cellLabel_, k)
    {
        if (cellLabel_[k] < 0)
        {
            FatalErrorIn("enhancedCloud::describeMesh")
                << "grid cell " << k << " has no OpenFOAM cell" << abort(FatalError);
        }
    }

    // cyclic patch pairs (the channel cases: blockMeshDict `cyclic`, in.lammps `boundary p f p`): smoothField couples
    // the first and the last cell layer along such an axis
    label cyc[3] = {0, 0, 0};
    const polyBoundaryMesh& patches = mesh_.boundaryMesh();
    forAll(patches, patchI)
    {
        const polyPatch& pp = patches[patchI];
        if (pp.type() == "cyclic" && pp.size() > 0)
        {
            const vector a = pp.faceAreas()[0];
            direction d = 0;
            if (mag(a.y()) > mag(a.component(d))) d = 1;
            if (mag(a.z()) > mag(a.component(d))) d = 2;
            cyc[d] = 1;
        }
    }
    for (direction d = 0; d < 3; d++)
    {
        reduce(cyc[d], maxOp<label>());
    }

    std::memset(&m, 0, sizeof(m));
    for (direction d = 0; d < 3; d++)
    {
        m.origin[d] = faces_[d][0];
        m.n[d] = n[d];
        m.dx[d] = (faces_[d][n[d]] - faces_[d][0])/n[d];
        // graded axis (blockMesh simpleGrading): the face coordinates; uniform axis: origin + i dx
        bool uniform = true;
        for (label i = 0; i < n[d]; i++)
        {
            if (mag((faces_[d][i + 1] - faces_[d][i]) - m.dx[d]) > 1e-9*m.dx[d]) uniform = false;
        }
        m.faces[d] = uniform ? NULL : faces_[d].cdata();
        m.periodic[d] = cyc[d];
    }
    m.cell_label = cellLabel_.cdata();
    m.slab_nx_global = 0;                      // every rank holds the whole mesh
}


// constant/cloudProperties + transportProperties -> the sf_cloud_props block; keys of softParticleCloud.C:433-486,
// enhancedCloud.C:543-608, createFields.H:126-157
void enhancedCloud::readProperties
(
    sf_cloud_props& pr,
    const IOdictionary& transDict,
    scalar diffusionBandWidth,
    label diffusionSteps
)
{
    std::memset(&pr, 0, sizeof(pr));
    const word model(cloudProperties_.lookup("dragModel"));          // newDragModel.C:38-41
    if (model == "ErgunWenYu") pr.dragModel = 0;
    else if (model == "SyamlalOBrien") pr.dragModel = 1;
    else if (model == "NoCorrection") pr.dragModel = 2;
    else
    {
        FatalErrorIn("enhancedCloud::readProperties")
            << "Unknown dragModel type " << model << nl << nl
            << "Valid dragModel types are :" << nl
            << "(ErgunWenYu NoCorrection SyamlalOBrien)" << abort(FatalError);
    }
    pr.subCycles = label(readScalar(cloudProperties_.lookup("subCycles")));
    pr.particleDrag = cloudProperties_.lookupOrDefault<Switch>("particleDrag", true);
    pr.particlePressureGrad = cloudProperties_.lookupOrDefault<Switch>("particlePressureGrad", true);
    pr.particleBuoyancy = cloudProperties_.lookupOrDefault<Switch>("particleBuoyancy", false);
    pr.particleAddedMass = cloudProperties_.lookupOrDefault<Switch>("particleAddedMass", false);
    pr.particleLift = cloudProperties_.lookupOrDefault<Switch>("particleLift", false);
    pr.particleHistoryForce = cloudProperties_.lookupOrDefault<Switch>("particleHistoryForce", false);
    pr.lubricationForce = cloudProperties_.lookupOrDefault<Switch>("lubricationForce", false);
    const vector g = cloudProperties_.lookupOrDefault<vector>("g", vector::zero);
    pr.gravity[0] = g.x(); pr.gravity[1] = g.y(); pr.gravity[2] = g.z();
    pr.rhob = dimensionedScalar(transDict.lookup("rhob")).value();
    pr.nub = dimensionedScalar(transDict.lookup("nub")).value();
    pr.maxPossibleAlpha = cloudProperties_.lookupOrDefault<scalar>("maxPossibleAlpha", 0.70);
    pr.diffusionBandWidth = diffusionBandWidth;
    pr.diffusionSteps = diffusionSteps;
    pr.UfSmooth = cloudProperties_.lookupOrDefault<Switch>("UfSmooth", true);
    pr.UpSmooth = cloudProperties_.lookupOrDefault<Switch>("UpSmooth", true);
    pr.dragSmooth = cloudProperties_.lookupOrDefault<Switch>("dragSmooth", true);
    pr.alphaSmooth = cloudProperties_.lookupOrDefault<Switch>("alphaSmooth", true);
    const tensor sd =
        cloudProperties_.lookupOrDefault<tensor>("smoothDirection", tensor(1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0));
    if (sd.xy() != 0 || sd.xz() != 0 || sd.yx() != 0 || sd.yz() != 0 || sd.zx() != 0 || sd.zy() != 0)
    {
        FatalErrorIn("enhancedCloud::readProperties")
            << "smoothDirection " << sd << ": only diagonal tensors are supported" << abort(FatalError);
    }
    pr.smoothDirection[0] = sd.xx(); pr.smoothDirection[1] = sd.yy(); pr.smoothDirection[2] = sd.zz();

    // the add / delete schedules are not part of the library; the inlet override of updateDragOnParticles is
    pr.addParticleOption = cloudProperties_.lookupOrDefault<label>("addParticle", 0);
    if
    (
        cloudProperties_.lookupOrDefault<label>("deleteParticle", 0) > 0
     || cloudProperties_.lookupOrDefault<label>("deleteBeforeAdd", 0) > 0
    )
    {
        FatalErrorIn("enhancedCloud::readProperties")
            << "deleteParticle / deleteBeforeAdd: particle delete schedules are not supported" << abort(FatalError);
    }
    if (pr.addParticleOption > 0)
    {
        Info<< "*** addParticle " << pr.addParticleOption
            << ": the inlet force override is applied, NO particles are added (schedule not supported)" << endl;
        const vector f = cloudProperties_.lookupOrDefault<vector>("inletForce", vector::zero);
        const vector e = cloudProperties_.lookupOrDefault<vector>("eccentricity", vector::zero);
        const tensor b = cloudProperties_.lookupOrDefault<tensor>("inletBox", tensor::zero);
        pr.inletForce[0] = f.x(); pr.inletForce[1] = f.y(); pr.inletForce[2] = f.z();
        pr.eccentricity[0] = e.x(); pr.eccentricity[1] = e.y(); pr.eccentricity[2] = e.z();
        for (direction k = 0; k < 9; k++)
        {
            pr.inletBox[k] = b.component(k);
        }
    }
}


void enhancedCloud::toGlobal(const vectorField& local, vectorField& global) const
{
    if (!Pstream::parRun())
    {
        global = local;
        return;
    }
    global.setSize(nCellsGlobal_);
    global = vector::zero;
    forAll(local, i)
    {
        global[cellOffset_ + i] = local[i];
    }
    Pstream::listCombineGather(global, plusEqOp<vector>());
    Pstream::listCombineScatter(global);
}


void enhancedCloud::toGlobal(const scalarField& local, scalarField& global) const
{
    if (!Pstream::parRun())
    {
        global = local;
        return;
    }
    global.setSize(nCellsGlobal_);
    global = 0.0;
    forAll(local, i)
    {
        global[cellOffset_ + i] = local[i];
    }
    Pstream::listCombineGather(global, plusEqOp<scalar>());
    Pstream::listCombineScatter(global);
}


void enhancedCloud::sumOverRanks(double* dev, label n) const
{
    if (!Pstream::parRun()) return;
    scalarField h(n);
    sfCheck(sf_dev_download(h.data(), dev, sizeof(double)*size_t(n), NULL), "enhancedCloud::sumOverRanks");
    Pstream::listCombineGather(h, plusEqOp<scalar>());
    Pstream::listCombineScatter(h);
    sfCheck(sf_dev_upload(dev, h.cdata(), sizeof(double)*size_t(n), NULL), "enhancedCloud::sumOverRanks");
}


void enhancedCloud::phase(int ph)
{
    sfCheck(sf_cloud_phase(cloud_, ph), "enhancedCloud::phase");
}


// Uf, DDtUf, grad p, curl Uf of the whole mesh in label order (updateDragOnParticles, enhancedCloud.C:112-116)
void enhancedCloud::setFluid()
{
    const vectorField gradp(fvc::grad(pf_)().internalField());
    const vectorField curlU(fvc::curl(Uf_)().internalField());
    vectorField gUf, gDDt, gGrad, gCurl;
    toGlobal(Uf_.internalField(), gUf);
    toGlobal(DDtUf_.internalField(), gDDt);
    toGlobal(gradp, gGrad);
    toGlobal(curlU, gCurl);
    // `vector` is three contiguous doubles (the reference relies on the same layout, softParticleCloud.C:823-846)
    sfCheck
    (
        sf_cloud_set_fluid
        (
            cloud_,
            reinterpret_cast<const double*>(gUf.cdata()), reinterpret_cast<const double*>(gDDt.cdata()),
            reinterpret_cast<const double*>(gGrad.cdata()), reinterpret_cast<const double*>(gCurl.cdata())
        ),
        "enhancedCloud::setFluid"
    );
}


void enhancedCloud::fetchAlphaUe()
{
    scalarField ga(nCellsGlobal_);
    vectorField gu(nCellsGlobal_);
    sfCheck(sf_cloud_get_fields(cloud_, ga.data(), reinterpret_cast<double*>(gu.data()), NULL, NULL), "enhancedCloud::fetchAlphaUe");
    scalarField& a = gamma_.internalField();
    vectorField& u = Ue_.internalField();
    forAll(a, i)
    {
        a[i] = ga[cellOffset_ + i];
        u[i] = gu[cellOffset_ + i];
    }
    gamma_.correctBoundaryConditions();
    Ue_.correctBoundaryConditions();
}


// the reference's timer buckets (writeCPUTime.H:1-19) from the library's
void enhancedCloud::updateTimers()
{
    sf_cloud_timers t;
    sfCheck(sf_cloud_get_timers(cloud_, &t), "enhancedCloud::updateTimers");
    cpuTimeSplit_[3] = t.dragOnParticles;    // foam -> lammps: drag closure + assembly straight into the fix fdrag rows
    cpuTimeSplit_[4] = t.lammps;             // lammps: the DEM sub-steps
    cpuTimeSplit_[5] = t.particleMove;       // lammps -> foam: cell owners of the new positions
    diffusionTimeCount_[0] = t.scatter;      // averaging + smoothing solves
    particleMoveTime_ = t.particleMove;
}

// * * * * * * * * * * * * * * * * Constructors  * * * * * * * * * * * * * * //

enhancedCloud::enhancedCloud
(
    const volVectorField& U,
    const volScalarField& p,
    volVectorField& Ue,
    const volVectorField& Uf,
    const volVectorField& DDtUf,
    dimensionedScalar nu,
    volScalarField& alpha,
    IOdictionary& cloudDict,
    IOdictionary& transDict,
    scalar diffusionBandWidth,
    label diffusionSteps
)
:
    lmp_(NULL),
    cloud_(NULL),
    mesh_(U.mesh()),
    runTime_(U.time()),
    pf_(p),
    Ue_(Ue),
    Uf_(Uf),
    DDtUf_(DDtUf),
    gamma_(alpha),
    Omega_
    (
        IOobject("Omega", U.time().timeName(), U.mesh(), IOobject::NO_READ, IOobject::AUTO_WRITE),
        U.mesh(),
        dimensionedScalar("zero", dimensionSet(1, -3, -1, 0, 0), scalar(0.0))
    ),
    Asrc_
    (
        IOobject("A_Source", U.time().timeName(), U.mesh(), IOobject::NO_READ, IOobject::AUTO_WRITE),
        U.mesh(),
        dimensionedVector("zero", dimensionSet(1, -2, -2, 0, 0), vector::zero),
        zeroGradientFvPatchVectorField::typeName
    ),
    cloudProperties_(cloudDict),
    nCellsGlobal_(0),
    cellOffset_(0),
    subCycles_(1),
    subSteps_(1),
    particleCount_(0),
    diffusionTimeCount_(2, 0.0),
    particleMoveTime_(0.0),
    cpuTimeSplit_(6, 0.0),
    dGamma_(NULL),
    dUe_(NULL),
    dAsrc_(NULL)
{
    initLammps();

    sf_cloud_mesh m;
    describeMesh(m);
    sf_cloud_props pr;
    readProperties(pr, transDict, diffusionBandWidth, diffusionSteps);

    // (adjustLampTimestep inside: softParticleCloud.C:209-261)
    sfCheck
    (
        sf_cloud_create(sedifoam_shim::h(lmp_), &m, &pr, runTime_.deltaTValue(), &cloud_),
        "enhancedCloud::enhancedCloud"
    );
    int sc = 1, ss = 1, nc = 0;
    sfCheck(sf_cloud_sub_cycling(cloud_, &sc, &ss), "enhancedCloud::enhancedCloud");
    subCycles_ = sc;
    subSteps_ = ss;
    sfCheck(sf_cloud_device_fields(cloud_, &dGamma_, &dUe_, &dAsrc_, &nc), "enhancedCloud::enhancedCloud");
    if (nc != nCellsGlobal_)
    {
        FatalErrorIn("enhancedCloud::enhancedCloud") << "cells " << nc << " " << nCellsGlobal_ << abort(FatalError);
    }
    particleCount_ = lammps_get_global_n(lmp_);

    // alpha and Ue of the initial particles (enhancedCloud.C:633: particleToEulerianField)
    if (Pstream::parRun())
    {
        // (sf_cloud_create scattered this rank's particles only)
        phase(2);
        sumOverRanks(dGamma_, nCellsGlobal_);
        sumOverRanks(dUe_, 3*nCellsGlobal_);
        phase(3);
    }
    fetchAlphaUe();

    // UfSmoothed of the initial condition (enhancedCloud.C:641-655)
    setFluid();
    phase(6);
    Info<< "initialization finished!" << endl;
}

// * * * * * * * * * * * * * * * * Destructor  * * * * * * * * * * * * * * * //

enhancedCloud::~enhancedCloud()
{
    if (cloud_) sf_cloud_destroy(cloud_);
    delete lmp_;          // softParticleCloud::finishLammps (softParticleCloud.C:354-366)
}

// * * * * * * * * * * * * * * * Member Functions  * * * * * * * * * * * * * //

void enhancedCloud::evolve()
{
    setFluid();
    if (!Pstream::parRun())
    {
        sfCheck(sf_cloud_evolve(cloud_), "enhancedCloud::evolve");
    }
    else
    {
        phase(0);                                    // next time step, UfSmoothed
        for (label k = 0; k < subCycles_; k++)
        {
            phase(1);                                // drag on this rank's particles -> fix fdrag rows
            lammps_step(lmp_, subSteps_);            // collective: halo, rebuild vote, migration inside the library
            if (k == 0)
            {
                phase(2);                            // per-cell sums of this rank's particles
                sumOverRanks(dGamma_, nCellsGlobal_);
                sumOverRanks(dUe_, 3*nCellsGlobal_);
                phase(3);                            // smoothing, Ue / gamma
            }
        }
    }
    fetchAlphaUe();
    particleCount_ = lammps_get_global_n(lmp_);
    updateTimers();
}


void enhancedCloud::calcTcFields()
{
    if (!Pstream::parRun())
    {
        sfCheck(sf_cloud_calc_tc_fields(cloud_), "enhancedCloud::calcTcFields");
    }
    else
    {
        phase(4);                                    // alpha cap, Asrc sums of this rank's particles
        sumOverRanks(dAsrc_, 3*nCellsGlobal_);
        phase(5);
    }
    vectorField gA(nCellsGlobal_);
    scalarField gO(nCellsGlobal_);
    sfCheck
    (
        sf_cloud_get_fields(cloud_, NULL, NULL, reinterpret_cast<double*>(gA.data()), gO.data()),
        "enhancedCloud::calcTcFields"
    );
    vectorField& A = Asrc_.internalField();
    scalarField& O = Omega_.internalField();
    forAll(A, i)
    {
        A[i] = gA[cellOffset_ + i];
        O[i] = gO[cellOffset_ + i];
    }
    Asrc_.correctBoundaryConditions();
    Omega_.correctBoundaryConditions();
    updateTimers();
}


void enhancedCloud::dragInfo()
{
    // sum of the particle drag per unit cell volume (enhancedCloud.C:1298-1338; "not parallel yet" there as well)
    const label n = sf_cloud_particle_count(cloud_);
    labelList cell(n);
    vectorField pDrag(n);
    sfCheck
    (
        sf_cloud_get_particles(cloud_, NULL, cell.data(), reinterpret_cast<double*>(pDrag.data()), NULL),
        "enhancedCloud::dragInfo"
    );
    vectorField sum(nCellsGlobal_, vector::zero);
    forAll(cell, i)
    {
        if (cell[i] >= 0) sum[cell[i]] += pDrag[i];
    }
    vectorField local(mesh_.nCells());
    forAll(local, i)
    {
        local[i] = sum[cellOffset_ + i]/mesh_.V()[i];
    }
    Info<< "Sum up drag on particle in unit cell: " << local << endl;
}


void enhancedCloud::averageInfo()
{
    double out[7];
    sfCheck(sf_cloud_average_info(cloud_, out), "enhancedCloud::averageInfo");
    scalar totalVolume = out[0];
    vector totalVel(out[1], out[2], out[3]);
    reduce(totalVel, sumOp<vector>());
    reduce(totalVolume, sumOp<scalar>());
    const vector averageVel = totalVel/(totalVolume + ROOTVSMALL);

    Info<< "total volume of particles is: " << totalVolume << endl;
    Info<< "total (velocity x volume) of particles is: " << totalVel << endl;
    Info<< "average velocity of all particles is: " << averageVel << endl;
}

} // End namespace Foam
