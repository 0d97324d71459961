// ErgunWenYuAmd.C -- ErgunWenYu::Jd (ErgunWenYu.C:86-145) through sfk_drag_model_jd(model 0).
#include "ErgunWenYuAmd.H"
#include "addToRunTimeSelectionTable.H"

#include "sedifoam_amd.h"

namespace Foam
{
    defineTypeNameAndDebug(ErgunWenYuAmd, 0);
    addToRunTimeSelectionTable(dragModel, ErgunWenYuAmd, dictionary);
}

Foam::ErgunWenYuAmd::ErgunWenYuAmd
(
    const dictionary& cloudDict,
    const IOdictionary& transDict,
    const scalarField& alpha,
    const scalarField& pd
)
:
    dragModel(cloudDict, transDict, alpha, pd),
    dUr_(NULL), dAlpha_(NULL), dPd_(NULL), dJd_(NULL), cap_(0)
{
    dimensionedScalar Dnuf_(transDict_.lookup("nub"));      // ErgunWenYu.C:65-69
    dimensionedScalar Drhof_(transDict_.lookup("rhob"));
    nuf_ = Dnuf_.value();
    rhof_ = Drhof_.value();
}

Foam::ErgunWenYuAmd::~ErgunWenYuAmd()
{
    sf_dev_free(dUr_); sf_dev_free(dAlpha_); sf_dev_free(dPd_); sf_dev_free(dJd_);
}

Foam::tmp<Foam::scalarField> Foam::ErgunWenYuAmd::Jd(const scalarField& Ur) const
{
    const label n = Ur.size();
    if (alpha_.size() != n || pd_.size() != n)               // ErgunWenYu.C:92-102
    {
        FatalErrorIn("ErgunWenYuAmd::Jd")
            << "size of Ur, alpha and pd differ: " << n << " " << alpha_.size() << " " << pd_.size()
            << abort(FatalError);
    }
    tmp<scalarField> tJd(new scalarField(n, 0.0));
    if (!n) return tJd;
    if (n > cap_)
    {
        sf_dev_free(dUr_); sf_dev_free(dAlpha_); sf_dev_free(dPd_); sf_dev_free(dJd_);
        cap_ = n + n/4 + 64;
        const size_t bytes = sizeof(scalar)*size_t(cap_);
        dUr_ = sf_dev_alloc(bytes); dAlpha_ = sf_dev_alloc(bytes); dPd_ = sf_dev_alloc(bytes); dJd_ = sf_dev_alloc(bytes);
    }
    const size_t bytes = sizeof(scalar)*size_t(n);
    sf_dev_upload(dUr_, Ur.cdata(), bytes, NULL);
    sf_dev_upload(dAlpha_, alpha_.cdata(), bytes, NULL);
    sf_dev_upload(dPd_, pd_.cdata(), bytes, NULL);
    if
    (
        sfk_drag_model_jd
        (
            0, n, static_cast<const double*>(dUr_), static_cast<const double*>(dAlpha_),
            static_cast<const double*>(dPd_), nuf_, rhof_, static_cast<double*>(dJd_), NULL
        ) != 0
    )
    {
        FatalErrorIn("ErgunWenYuAmd::Jd") << sf_last_error() << abort(FatalError);
    }
    sf_dev_download(tJd().data(), dJd_, bytes, NULL);
    return tJd;
}
