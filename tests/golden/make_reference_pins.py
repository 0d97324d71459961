#!/usr/bin/env python3
"""Golden vectors computed by the REFERENCE'S OWN SOURCE LINES for the closures its test cases never exercise
(gran/hertzFix/history, fix cohesive, fix fdrag, ErgunWenYu, SyamlalOBrien): tests/golden/reference_pins.json.

Run in the build container only (needs /root/reference; nothing of it is copied: the JSON holds numbers).

LAMMPS and OpenFOAM are not installed, so the reference's translation units cannot be compiled here -- but the arithmetic
of the hot path sits in a handful of plain C loops.  This script reads those line ranges from the reference files AT
RUN TIME, transliterates them statement by statement into Python (same operators, same order; `sqrt`, `log`, `exp`,
`pow`, `atan` are libm's through the `math` module; IEEE doubles, no contraction) and executes them on seeded inputs laid
out like LAMMPS' arrays (x[i][k], firstneigh[i][jj], firstshear[i][3 jj], ...).  What comes out is what the reference's
lines compute, not what anybody's restatement of them computes: tests/test_reference_pins.py holds the oracle (and through
it the HIP path) to these numbers bit for bit (a few ulp where the oracle shares a sub-expression).

  pair_gran_hertzFix_history.cpp:109-286   the ii / jj loop of PairGranHertzFixHistory::compute
  fix_cohesive.cpp:161-262                 FixCohe::post_force, opt 0 and opt 1
  fix_fluid_drag.cpp:143-163               FixFluidDrag::post_force
  pair_lubricate_poly.cpp:193-407 + :539-559   PairLubricatePoly::compute's loop and init_style's constants
  fix_wall_granFix.cpp:286-344 + :361-436, :446-553, :563-678   FixWallGranFix::post_force and its three contact laws
  ErgunWenYu.C:104-132, SyamlalOBrien.C:105-143   dragModel::Jd (OpenFOAM field algebra, evaluated element by element)
"""
import json
import math
import os
import random
import re
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

PY_KEYWORDS = {"del", "lambda", "in", "is", "not", "and", "or", "from", "pass", "def", "class", "global", "with", "as",
               "yield", "try", "except", "raise", "import", "None", "True", "False", "print", "exec", "assert", "type"}
TYPE_WORDS = r"(?:const\s+)?(?:unsigned\s+)?(?:tmp\s*<\s*scalarField\s*>|(?:scalarField|vectorField|double|int|float|scalar|label|bool|vector|softParticle)\b)"


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    # preprocessor blocks (#ifdef DEBUG ... #endif) hold only printouts in these ranges
    out, skip = [], 0
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#if"):
            skip += 1
            continue
        if s.startswith("#endif"):
            skip -= 1
            continue
        if s.startswith("#") or skip:
            continue
        out.append(line)
    return "\n".join(out)


class Parser:
    """statement-level parser of the C subset the ranges use: blocks, if / else (braced or not), counting for loops,
    forAll, expression statements (with comma sequences and `a = b = c` chains), declarations"""

    def __init__(self, src):
        self.s = src
        self.p = 0

    def ws(self):
        while self.p < len(self.s) and self.s[self.p].isspace():
            self.p += 1

    def word(self):
        m = re.compile(r"[A-Za-z_]\w*").match(self.s, self.p)
        return m.group(0) if m else None

    def parens(self):
        self.ws()
        assert self.s[self.p] == "(", self.s[self.p:self.p + 40]
        depth, q = 0, self.p
        while True:
            c = self.s[q]
            depth += c == "("
            depth -= c == ")"
            q += 1
            if depth == 0:
                break
        inner = self.s[self.p + 1:q - 1]
        self.p = q
        return inner

    def block(self, ind):
        self.ws()
        assert self.s[self.p] == "{"
        self.p += 1
        out = []
        while True:
            self.ws()
            if self.s[self.p] == "}":
                self.p += 1
                break
            out += self.stmt(ind)
        return out or [ind + "pass"]

    def stmt(self, ind):
        self.ws()
        if self.p >= len(self.s):
            return []
        if self.s[self.p] == "{":
            return self.block(ind)
        if self.s[self.p] == ";":
            self.p += 1
            return []
        w = self.word()
        if w == "if":
            self.p += 2
            cond = expr(self.parens())
            body = self.stmt(ind + "    ") or [ind + "    pass"]
            out = [ind + "if " + cond + ":"] + body
            self.ws()
            if self.word() == "else":
                self.p += 4
                self.ws()
                if self.word() == "if":
                    rest = self.stmt(ind)
                    rest[0] = ind + "el" + rest[0].strip()
                    out += rest
                else:
                    out += [ind + "else:"] + (self.stmt(ind + "    ") or [ind + "    pass"])
            return out
        if w == "for":
            self.p += 3
            init, cond, inc = [t.strip() for t in self.parens().split(";")]
            init = re.sub(r"^" + TYPE_WORDS + r"\s+", "", init)
            m1 = re.match(r"(\w+)\s*=\s*(.+)$", init)
            m2 = re.match(r"(\w+)\s*<\s*(.+)$", cond)
            assert m1 and m2 and m1.group(1) == m2.group(1) and re.match(m1.group(1) + r"\s*\+\+$", inc), (init, cond, inc)
            body = self.stmt(ind + "    ") or [ind + "    pass"]
            return [ind + "for %s in range(%s, %s):" % (m1.group(1), expr(m1.group(2)), expr(m2.group(2)))] + body
        if w == "forAll":
            self.p += 6
            lst, idx = [t.strip() for t in self.parens().split(",")]
            body = self.stmt(ind + "    ") or [ind + "    pass"]
            return [ind + "for %s in range(len(%s)):" % (idx, ident(lst))] + body
        if w in ("continue", "break"):
            self.p += len(w)
            self.ws()
            assert self.s[self.p] == ";"
            self.p += 1
            return [ind + w]
        if w == "return":
            q = self.s.index(";", self.p)
            e = self.s[self.p + 6:q]
            self.p = q + 1
            return [ind + "_result = " + expr(e)]
        # expression statement up to the ';' at depth 0
        depth, q = 0, self.p
        while not (self.s[q] == ";" and depth == 0):
            depth += self.s[q] in "(["
            depth -= self.s[q] in ")]"
            q += 1
        text = " ".join(self.s[self.p:q].split())
        self.p = q + 1
        return [ind + t for t in expr_stmt(text)]


def ident(name):
    return name + "_" if name in PY_KEYWORDS else name


def expr(e):
    e = " ".join(e.split())
    e = re.sub(r"&\s*(\w+)\s*\[([^\]]+)\]", r"_View(\1, \2)", e)          # shear = &allshear[3*jj]
    e = e.replace("->", ".")
    e = re.sub(r"(?<![\w.])(\d+)\.(?![\d\w])", r"\1.0", e)                # `0.` -> `0.0`
    e = re.sub(r"(?<![\d.])[A-Za-z_]\w*", lambda m: ident(m.group(0)), e)    # (not the e of 1e-3)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    e = re.sub(r"\bscalar\s*\(", "float(", e)
    return e.strip()


def split_top(text, sep):
    parts, depth, cur = [], 0, ""
    for c in text:
        depth += c in "(["
        depth -= c in ")]"
        if c == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += c
    parts.append(cur)
    return [p.strip() for p in parts]


def expr_stmt(text):
    m = re.match(r"^(" + TYPE_WORDS + r")\s*&?\s*(.*)$", text)
    decl = bool(m)
    if decl:
        text = m.group(2)
    out = []
    for part in split_top(text, ","):        # declarator list, or a C comma sequence of assignments
        part = part.lstrip("*& ").strip()
        mc = re.match(r"^(\w+)\s*\((.*)\)$", part) if decl else None
        if mc and not re.search(r"(?<![=!<>+\-*/])=(?!=)", part):
            out.append("%s = %s" % (ident(mc.group(1)), expr(mc.group(2))))   # `vector FH(vector::zero);`
            continue
        if decl and not re.search(r"(?<![=!<>+\-*/])=(?!=)", part):
            continue                          # `double a, b;`
        mi = re.match(r"^(\w+)\s*(\+\+|--)$", part)
        if mi:
            out.append("%s %s= 1" % (ident(mi.group(1)), mi.group(2)[0]))
            continue
        out.append(expr(part))
    return out


# sha256 of every line range this script transliterates and executes: the reference is untrusted input, and a range that
# drifted upstream would silently pin other code.  A mismatch stops the run (--rehash prints the table of the tree as it is).
RANGE_SHA256 = {
    "interfaceToLammps/fix_cohesive.cpp:161-262": "003e983519653da296838b2904397a4e269bd878e125abaa75baa1112c34644f",
    "interfaceToLammps/fix_fluid_drag.cpp:143-163": "83502a041fa38900270b79fc0904ed49adfe15229c6e1a76e880ad342c72ef7b",
    "interfaceToLammps/fix_wall_granFix.cpp:286-344": "d245fef1bd51f68b65958a0f84181ef38a2976abca9aa15273c9574424018b92",
    "interfaceToLammps/fix_wall_granFix.cpp:361-436": "3cae0ab1a03d4fe9f22aa64f1baa84a115f3d0079f9850342abfa296505eba6f",
    "interfaceToLammps/fix_wall_granFix.cpp:446-553": "308811f0d5eb8203f64e20d27398b357a607cf0baf3597938d15cdbedf151c27",
    "interfaceToLammps/fix_wall_granFix.cpp:563-678": "9c1234ff74feb4f6d4ed7d48304c104c6fc618711e36a018d939d1f1da71a6d9",
    "interfaceToLammps/pair_gran_hertzFix_history.cpp:109-286": "2d86a6765e5bf2d008296f938bfba0132fa7bfad6e9bd4a3ad699a46093f3f62",
    "interfaceToLammps/pair_lubricate_poly.cpp:193-407": "f1aaf4d7ad20d6bd0347dfb98af97e235f8b3b1c6d0f233af8f18cef1703547c",
    "interfaceToLammps/pair_lubricate_poly.cpp:539-559": "9a7d10bfec50de372b8ee8f2249640ae499d124644ecc5dafea34ae40ca78cb1",
    "lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C:104-132": "e5d4f42819875f2598b9c41b7a8ff017ba90da8b62d776d04f8f085067d35445",
    "lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C:105-143": "afe1a4b13462921569630cfd38a6a48c2d1f548405dda559d032d7e4b3f1ebb5",
    "lammpsFoam/enhancedCloud.C:129-311": "5d5af2bcdf07ed7d45efdc1d8a59482ee731e89b382f5777ce3e6c63ba72a8a9",
    "lammpsFoam/enhancedCloud.C:1374-1383": "04e9b7228a0e9277c73347e6441d2c40a313df84b2a34624adfc5f7078536805",
    "lammpsFoam/enhancedCloud.C:318-439": "6ef6f1f544d1645f54d08146bde67552cfaa15f2135db1fdebbbc67077def731",
    "lammpsFoam/enhancedCloud.C:41-51": "7b080873d889934e87dd0eefe18fc348f76420a3ffac20aca3226a2ae68b2c80",
    "lammpsFoam/enhancedCloud.C:59-75": "b0f6850a59cbf579c30c7fe77038b905a573dfe1117210dd76e1f0c1581d482a",
    "lammpsFoam/enhancedCloud.C:88-108": "51ebac1e81262371b0cb9374f25cc4f18570bde0ca2bc78c675bb2f10025f3db",
    "lammpsFoam/enhancedCloud.C:913-979": "0c3d67732147530f150d4cf566da1823a1365093b998ef0c69aab9c4eda6a043",
    "lammpsFoam/softParticle.C:72-72": "e6aa5f11acf47d49b473f1b0bb265c022edab593223d53710ce773fc839f36db",
    "lammpsFoam/softParticle.H:272-272": "7c76c48bc2e1e9280d76584046263afd261c00c432ee53082a42d6ea783c197d",
    "lammpsFoam/softParticleCloud.C:1356-1416": "7251901c4d147f687814a665419c5f136327c426f8d35a82486cf6bac884e8bb",
}
SAFE_BUILTINS = {"range": range, "len": len, "abs": abs, "float": float, "int": int, "min": min, "max": max, "bool": bool}


def run(code, ns):
    """execute transliterated reference lines with nothing but arithmetic in reach: no import, no open, no eval"""
    if re.search(r"__|\bimport\b|\bopen\b|\beval\b|\bexec\b|\bcompile\b|\bgetattr\b|\bglobals\b|\blocals\b", code):
        raise SystemExit("make_reference_pins: the transliterated code holds a name it must not:\n" + code[:400])
    ns["__builtins__"] = SAFE_BUILTINS
    exec(code, ns)


def translate(path, first, last, dialect=None):
    lines = open(os.path.join(REF, path)).read().split("\n")[first - 1:last]
    import hashlib
    key = "%s:%d-%d" % (path, first, last)
    digest = hashlib.sha256("\n".join(lines).encode()).hexdigest()
    if "--rehash" in sys.argv:
        print('    "%s": "%s",' % (key, digest))
    elif RANGE_SHA256.get(key) != digest:
        raise SystemExit("make_reference_pins: %s is not the text this script was written against (sha256 %s, expected %s)"
                         % (key, digest, RANGE_SHA256.get(key)))
    src = strip_comments("\n".join(lines))
    if dialect:
        src = dialect(src)
    P = Parser(src)
    out = []
    while True:
        P.ws()
        if P.p >= len(src):
            break
        out += P.stmt("")
    return "\n".join(out)


class _View:
    """a C pointer into an array: shear = &allshear[3*jj]"""

    def __init__(self, a, off):
        self.a, self.off = a, off

    def __getitem__(self, k):
        return self.a[self.off + k]

    def __setitem__(self, k, v):
        self.a[self.off + k] = v


MATH = dict(sqrt=math.sqrt, log=math.log, exp=math.exp, pow=math.pow, fabs=math.fabs, atan=math.atan, _View=_View,
            printf=lambda *a: None, MY_PI=3.14159265358979323846,   # [3P] math_const.h
            MIN=lambda a, b: a if a < b else b, MAX=lambda a, b: a if a > b else b)   # [3P] lmptype.h macros


def hexs(a):
    return [float(v).hex() for v in a]


# ---------------------------------------------------------------------------------------------------------------------
def cluster(rng, n, d0, poly):
    """n spheres in a small cloud: about half of the listed pairs overlap"""
    side = max(2, int(round(n ** (1.0 / 3.0))))
    x, r = [], []
    for k in range(n):
        ix, iy, iz = k % side, (k // side) % side, k // (side * side)
        rad = 0.5 * d0 * (rng.uniform(0.6, 1.4) if poly else 1.0)
        x.append([d0 * (0.93 * ix + rng.uniform(-0.12, 0.12)), d0 * (0.93 * iy + rng.uniform(-0.12, 0.12)),
                  d0 * (0.93 * iz + rng.uniform(-0.12, 0.12))])
        r.append(rad)
    return x, r


def half_list(x, r, extra, nlocal):
    first = [[] for _ in range(len(x))]
    for i in range(nlocal):
        for j in range(i + 1, len(x)):
            d2 = sum((x[i][k] - x[j][k]) ** 2 for k in range(3))
            if d2 < (r[i] + r[j] + extra) ** 2:
                first[i].append(j)
    return first


def case_hertz(code, seed, n, nlocal, poly, shearupdate, frozen, rigid=False):
    rng = random.Random(seed)
    d0 = 1.0e-3
    x, radius = cluster(rng, n, d0, poly)
    v = [[rng.uniform(-0.3, 0.3) for _ in range(3)] for _ in range(n)]
    omega = [[rng.uniform(-200.0, 200.0) for _ in range(3)] for _ in range(n)]
    rmass = [2650.0 * 4.0 / 3.0 * math.pi * rr ** 3 for rr in radius]
    mask = [1 | (2 if (frozen and rng.random() < 0.2) else 0) for _ in range(n)]
    firstneigh = half_list(x, radius, 0.15 * d0, nlocal)
    numneigh = [len(l) for l in firstneigh]
    firsttouch = [[(1 if rng.random() < 0.6 else 0) for _ in l] for l in firstneigh]
    firstshear = [[(rng.uniform(-2e-6, 2e-6) if firsttouch[i][jj // 3] else 0.0) for jj in range(3 * len(l))]
                  for i, l in enumerate(firstneigh)]
    inp = dict(n=n, nlocal=nlocal, x=x, v=v, omega=omega, radius=radius, rmass=rmass, mask=mask, firstneigh=firstneigh,
               touch=[list(t) for t in firsttouch], shear=[list(s) for s in firstshear], kn=1.0e7, kt=2.0e7 / 7.0,
               gamman=0.5, xmu=0.4, dt=1.0e-6, shearupdate=shearupdate, freeze_group_bit=2 if frozen else 0)
    # the fix_rigid branch (:182-185): a third of the atoms belong to rigid bodies and collide with their body's mass
    mass_rigid = None
    if rigid:
        mass_rigid = [(rng.uniform(5.0, 40.0) * rmass[k] if rng.random() < 0.35 else 0.0) for k in range(n)]
        inp["mass_rigid"] = mass_rigid
    f = [[0.0] * 3 for _ in range(n)]
    torque = [[0.0] * 3 for _ in range(n)]
    ns = dict(MATH, inum=nlocal, ilist=list(range(nlocal)), x=x, v=v, omega=omega, radius=radius, rmass=rmass, mass=None,
              type_=[1] * n, mask=mask, freeze_group_bit=inp["freeze_group_bit"], fix_rigid=1 if rigid else 0,
              mass_rigid=mass_rigid,
              firstneigh=firstneigh, numneigh=numneigh, firsttouch=firsttouch, firstshear=firstshear,
              NEIGHMASK=0x3FFFFFFF, kn=inp["kn"], kt=inp["kt"], gamman=inp["gamman"], xmu=inp["xmu"], dt=inp["dt"],
              shearupdate=shearupdate, nlocal=nlocal, evflag=0, ev_tally_xyz=lambda *a: None, f=f, torque=torque)
    run(code, ns)
    out = dict(f=[hexs(a) for a in f], torque=[hexs(a) for a in torque], touch=firsttouch,
               shear=[hexs(a) for a in firstshear])
    return dict(inp=inp, out=out)


def case_cohesive(code, seed, n, nlocal, opt, newton_pair, smin, group):
    rng = random.Random(seed)
    d0 = 1.0e-4
    x, radius = cluster(rng, n, d0, True)
    # a few pairs at hand-set gaps so that every branch is taken: del > lam/pi, smin < del <= lam/pi, del <= smin, overlap
    lam = 1.0e-7
    for k, gap in enumerate((5.0e-7, 2.0e-8, 0.3 * smin, -1.0e-8)):
        i, j = 2 * k, 2 * k + 1
        x[j] = [x[i][0] + radius[i] + radius[j] + gap, x[i][1], x[i][2]]
    mask = [1 | (4 if (not group or rng.random() < 0.7) else 0) for _ in range(n)]
    smax = 0.1 * d0
    firstneigh = half_list(x, radius, 1.5 * smax, nlocal)
    f = [[0.0] * 3 for _ in range(n)]
    inp = dict(n=n, nlocal=nlocal, x=x, radius=radius, mask=mask, firstneigh=firstneigh, ah=1.0e-20, lam=lam, smin=smin,
               smax=smax, opt=opt, newton_pair=newton_pair, groupbit=4)

    class Err:
        def all(self, *a):
            raise RuntimeError("invalid option")
    ns = dict(MATH, opt=opt, inum=nlocal, nlocal=nlocal, ilist=list(range(nlocal)), mask=mask, groupbit=4, x=x,
              radius=radius, firstneigh=firstneigh, numneigh=[len(l) for l in firstneigh], smax=smax, smin=smin, lam=lam,
              ah=inp["ah"], PInv=0.25 / math.atan(1.0), newton_pair=newton_pair, f=f, error=Err(), FLERR=0)
    run(code, ns)
    return dict(inp=inp, out=dict(f=[hexs(a) for a in f]))


def case_fdrag(code, seed, n, carrier_rho):
    rng = random.Random(seed)
    radius = [0.5e-3 * rng.uniform(0.5, 1.5) for _ in range(n)]
    rmass = [2650.0 * 4.0 / 3.0 * math.pi * r ** 3 for r in radius]
    v = [[rng.uniform(-0.1, 0.1) for _ in range(3)] for _ in range(n)]
    vOld = [[rng.uniform(-0.1, 0.1) for _ in range(3)] for _ in range(n)]
    ffl = [[rng.uniform(-1e-6, 1e-6) for _ in range(3)] for _ in range(n)]
    DuDt = [[rng.uniform(-5.0, 5.0) for _ in range(3)] for _ in range(n)]
    mask = [1 | (2 if rng.random() < 0.8 else 0) for _ in range(n)]
    f = [[rng.uniform(-1e-6, 1e-6) for _ in range(3)] for _ in range(n)]
    inp = dict(n=n, radius=radius, rmass=rmass, v=v, vOld=[list(a) for a in vOld], ffluiddrag=ffl, DuDt=DuDt, mask=mask,
               groupbit=2, dt=1.0e-6, carrier_rho=carrier_rho, f=[list(a) for a in f])
    ns = dict(MATH, nlocal=n, mask=mask, groupbit=2, rmass=rmass, r=radius, v=v, vOld=vOld, timeStep=1.0e-6,
              ffluiddrag=ffl, DuDt=DuDt, carrier_rho=carrier_rho, f=f, rho=0.0, accX=0.0, accY=0.0, accZ=0.0)
    run(code, ns)
    return dict(inp=inp, out=dict(f=[hexs(a) for a in f], vOld=[hexs(a) for a in vOld]))


def make_function(name, args, path, first, last, ns):
    """a C function body (the lines after its declarations) as a Python function in namespace ns; the members it reads
    (kn, kt, gamman, gammat, xmu, dt, shearupdate ...) are looked up there"""
    body = translate(path, first, last)
    src = "def %s(%s):\n" % (name, ", ".join(args)) + "\n".join("    " + l for l in body.split("\n"))
    run(src, ns)
    return src


def case_wall(seed, n, pairstyle, wallstyle, shearupdate):
    """FixWallGranFix::post_force (fix_wall_granFix.cpp:286-344) with its three laws (:347-437 hooke, :441-554
    hooke_history, :558-679 hertz_history) on atoms between two plane walls"""
    rng = random.Random(seed)
    path = "interfaceToLammps/fix_wall_granFix.cpp"
    d0 = 1.0e-3
    lo, hi = 0.0, 8.0 * d0
    radius = [0.5 * d0 * rng.uniform(0.7, 1.3) for _ in range(n)]
    x = [[rng.uniform(0.0, 8.0 * d0) for _ in range(3)] for _ in range(n)]
    for i in range(n):    # two thirds of the atoms overlap one of the two walls
        if i % 3 == 0:
            x[i][wallstyle] = lo + radius[i] * rng.uniform(0.80, 0.999)
        elif i % 3 == 1:
            x[i][wallstyle] = hi - radius[i] * rng.uniform(0.80, 0.999)
        else:
            x[i][wallstyle] = rng.uniform(lo + 1.5 * d0, hi - 1.5 * d0)
    v = [[rng.uniform(-0.3, 0.3) for _ in range(3)] for _ in range(n)]
    omega = [[rng.uniform(-200.0, 200.0) for _ in range(3)] for _ in range(n)]
    rmass = [2650.0 * 4.0 / 3.0 * math.pi * r ** 3 for r in radius]
    mask = [1 | (2 if rng.random() < 0.85 else 0) for _ in range(n)]
    shear = [[rng.uniform(-2e-6, 2e-6) for _ in range(3)] for _ in range(n)]
    f = [[0.0] * 3 for _ in range(n)]
    torque = [[0.0] * 3 for _ in range(n)]
    inp = dict(n=n, pairstyle=pairstyle, wallstyle=wallstyle, lo=lo, hi=hi, x=x, v=v, omega=omega, radius=radius,
               rmass=rmass, mask=mask, groupbit=2, shear=[list(a) for a in shear], kn=1.0e7, kt=2.0e7 / 7.0, gamman=0.5,
               gammat=0.25, xmu=0.4, dt=1.0e-6, shearupdate=shearupdate)
    ns = dict(MATH, XPLANE=0, YPLANE=1, ZPLANE=2, ZCYLINDER=3, HOOKE=0, HOOKE_HISTORY=1, HERTZ_HISTORY=2,
              kn=inp["kn"], kt=inp["kt"], gamman=inp["gamman"], gammat=inp["gammat"], xmu=inp["xmu"], dt=inp["dt"],
              shearupdate=shearupdate, wallstyle=wallstyle, pairstyle=pairstyle, wlo=lo, whi=hi, wshear=0, axis=0,
              vshear=0.0, cylradius=0.0, vwall=[0.0, 0.0, 0.0], nlocal=n, mask=mask, groupbit=2, x=x, v=v, f=f,
              omega=omega, torque=torque, radius=radius, rmass=rmass, shear=shear)
    args = ["rsq", "dx", "dy", "dz", "vwall", "v", "f", "omega", "torque", "radius", "mass"]
    make_function("hooke", args, path, 361, 436, ns)
    make_function("hooke_history", args + ["shear"], path, 446, 553, ns)
    make_function("hertz_history", args + ["shear"], path, 563, 678, ns)
    ns["update"] = type("Update", (), {"setupflag": 0 if shearupdate else 1})()   # :286-287 derive shearupdate from it
    ns["shearupdate"] = -1
    run(translate(path, 286, 344), ns)
    return dict(inp=inp, out=dict(f=[hexs(a) for a in f], torque=[hexs(a) for a in torque], shear=[hexs(a) for a in shear]))


def case_lubricate(seed, n, nlocal, flaglog, flagfld, flagVF, cut_inner_d=1.45):
    """PairLubricatePoly: the volume-fraction constants of init_style (pair_lubricate_poly.cpp:539-559, the MPI_Allreduce
    of one rank being the identity) and the ii / jj loop of compute (:193-407) on a full list"""
    rng = random.Random(seed)
    path = "interfaceToLammps/pair_lubricate_poly.cpp"
    d0 = 1.0e-3
    x, radius = cluster(rng, n, d0, True)
    for i in range(n):   # a looser cloud: gaps of 0 .. 0.3 d between the surfaces, a few overlaps
        x[i] = [c * 1.18 for c in x[i]]
    v = [[rng.uniform(-0.3, 0.3) for _ in range(3)] for _ in range(n)]
    omega = [[rng.uniform(-200.0, 200.0) for _ in range(3)] for _ in range(n)]
    # (1.45 d: an inner cutoff above the largest ri + rj, no log of a negative gap; the last case puts it BELOW the contact
    # distance of most pairs: an overlapping pair beyond it keeps h_sep < 0, :286-300, and C's log returns NaN)
    mu, cut_inner, cut_global = 1.0e-3, cut_inner_d * d0, 1.9 * d0
    vol_T = (1.18 * d0 * 4.0) ** 3
    full = [[j for j in range(n) if j != i and sum((x[i][k] - x[j][k]) ** 2 for k in range(3)) < (1.1 * cut_global) ** 2]
            for i in range(nlocal)]
    inp = dict(n=n, nlocal=nlocal, x=x, v=v, omega=omega, radius=radius, firstneigh=full, mu=mu, flaglog=flaglog,
               flagfld=flagfld, flagHI=1, flagVF=flagVF, cut_inner=cut_inner, cut_global=cut_global, vol_T=vol_T)
    atom = type("Atom", (), {"radius": radius})()

    def allreduce(src, dst, *a):
        dst[0] = src[0]
    ns = dict(MATH, nlocal=n, atom=atom, vol_T=vol_T, flagVF=flagVF, flaglog=flaglog, mu=mu,
              MPI_Allreduce=allreduce, MPI_DOUBLE=0, MPI_SUM=0, world=0)
    # :539-559 -- `MPI_Allreduce(&volP,&vol_P,...)` with one rank: vol_P = volP
    code = re.sub(r"MPI_Allreduce\(&volP,&vol_P,[^\n]*", "vol_P = volP", translate(path, 539, 559))
    run(code, ns)
    R0, RT0, RS0 = ns["R0"], ns["RT0"], ns["RS0"]
    f = [[0.0] * 3 for _ in range(n)]
    torque = [[0.0] * 3 for _ in range(n)]
    z3 = lambda: [0.0, 0.0, 0.0]
    def c_log(a):   # libm: log(negative) = NaN, log(0) = -inf (math.log raises instead)
        return math.log(a) if a > 0.0 else (-math.inf if a == 0.0 else math.nan)
    ns = dict(MATH, log=c_log, inum=nlocal, ilist=list(range(nlocal)), x=x, v=v, omega=omega, radius=radius, atom=atom,
              type_=[1] * n, firstneigh=full, numneigh=[len(l) for l in full], flagfld=flagfld, flagHI=1, flaglog=flaglog,
              vxmu2f=1.0, R0=R0, RT0=RT0, RS0=RS0, mu=mu, shearing=0, vflag_either=0, evflag=0, nlocal=nlocal,
              newton_pair=0, Ef=[z3(), z3(), z3()], cutsq=[[0.0, 0.0], [0.0, cut_global * cut_global]],
              cut_inner=[[0.0, 0.0], [0.0, cut_inner]], wi=z3(), wj=z3(), xl=z3(), jl=z3(), vi=z3(), vj=z3(), overlaps=0,
              f=f, torque=torque, ev_tally_xyz=lambda *a: None, v_tally_tensor=lambda *a: None)
    run(translate(path, 193, 407), ns)
    return dict(inp=inp, out=dict(R0=float(R0).hex(), RT0=float(RT0).hex(), RS0=float(RS0).hex(),
                                  f=[hexs(a) for a in f], torque=[hexs(a) for a in torque], overlaps=ns["overlaps"]))


class Field(list):
    """an OpenFOAM scalarField of ONE element evaluated with Python floats (libm pow / sqrt, IEEE + - * /)"""

    def _b(self, o, op):
        return Field([op(a, (o[k] if isinstance(o, Field) else o)) for k, a in enumerate(self)])

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: b + a)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: b - a)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._b(o, lambda a, b: b / a)
    def __call__(self): return self            # tmp<scalarField>::operator()


def fmax(a, b):
    return a._b(b, max) if isinstance(a, Field) else (b._b(a, max) if isinstance(b, Field) else max(a, b))


def fpow(a, b):
    return a._b(b, math.pow) if isinstance(a, Field) else math.pow(a, b)


def fsqrt(a):
    return Field([math.sqrt(v) for v in a]) if isinstance(a, Field) else math.sqrt(a)


def fsqr(a):
    return a * a


def case_drag(code, result_name, seed, n, nuf, rhof):
    rng = random.Random(seed)
    Ur = [10.0 ** rng.uniform(-5.0, 1.0) for _ in range(n)]
    alpha = [rng.choice([0.0, 1.0e-3, 0.1, 0.19, 0.2, 0.21, 0.4, 0.6, 0.64, rng.uniform(0.0, 0.7)]) for _ in range(n)]
    pd = [10.0 ** rng.uniform(-4.3, -2.3) for _ in range(n)]
    Ur[0], pd[0], alpha[0] = 30.0, 5.0e-3, 0.05       # Re > 1000
    Ur[1] = 0.0                                        # Re clamps to ROOTVSMALL
    out = []
    for k in range(n):
        ns = dict(max=fmax, pow=fpow, sqrt=fsqrt, sqr=fsqr, float=float, len=len, range=range, ROOTVSMALL=1.0e-150,
                  alpha_=Field([alpha[k]]), pd_=Field([pd[k]]), Ur=Field([Ur[k]]), nuf_=nuf, rhof_=rhof)
        run(code, ns)
        out.append(ns[result_name][0])
    return dict(inp=dict(n=n, Ur=Ur, alpha=alpha, pd=pd, nuf=nuf, rhof=rhof), out=dict(Jd=hexs(out)))



# ---------------------------------------------------------------------------------------------------------------------
# The OpenFOAM-side loops of enhancedCloud.C (drag assembly :129-311 with g1n :1374-1383 and pointInRegion
# softParticleCloud.C:1356-1416; particleToEulerianField :913-979; calcTcFields :318-439): plain loops over `vector`s and
# particle accessors.  A three-component Vec with OpenFOAM's operator set (+ - * / ^ & mag, evaluated component by component in
# OpenFOAM's order), list-backed fields and a particle object with the accessors of softParticle.H let the same
# statement-by-statement transliteration run them.

class Vec:
    __slots__ = ("a",)

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.a = (float(x), float(y), float(z))

    def x(self): return self.a[0]
    def y(self): return self.a[1]
    def z(self): return self.a[2]
    def component(self, k): return self.a[k]
    def __add__(self, o): return Vec(self.a[0] + o.a[0], self.a[1] + o.a[1], self.a[2] + o.a[2])
    def __sub__(self, o): return Vec(self.a[0] - o.a[0], self.a[1] - o.a[1], self.a[2] - o.a[2])
    def __neg__(self): return Vec(-self.a[0], -self.a[1], -self.a[2])
    def __mul__(self, s): return Vec(self.a[0] * s, self.a[1] * s, self.a[2] * s)        # vs * s   (VectorSpaceI.H)
    def __rmul__(self, s): return Vec(s * self.a[0], s * self.a[1], s * self.a[2])       # s * vs
    def __truediv__(self, s): return Vec(self.a[0] / s, self.a[1] / s, self.a[2] / s)    # vs / s: component / s
    def __xor__(self, o):                                                                 # cross product (VectorI.H)
        a, b = self.a, o.a
        return Vec(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
    def __and__(self, o): return self.a[0] * o.a[0] + self.a[1] * o.a[1] + self.a[2] * o.a[2]   # inner product


def fmag(v):
    if isinstance(v, Vec):   # sqrt(magSqr): x*x, += y*y, += z*z
        return math.sqrt(v.a[0] * v.a[0] + v.a[1] * v.a[1] + v.a[2] * v.a[2])
    return abs(v)


class Tensor9(list):
    def component(self, k): return self[k]


class GeoField:
    """a vol*Field seen through what the loops touch: internalField() (here .f), operator[], oldTime()"""

    def __init__(self, values, old=None):
        self.f = Field(values)
        self.old = old

    def __getitem__(self, k): return self.f[k]
    def oldTime(self): return self.old
    def correctBoundaryConditions(self): pass


class Particle:
    """softParticle.H:196-290: the accessors the loops call; Vol() and the mass are the reference's own lines"""

    def __init__(self, ns, pos, U, UOld, d, rho, cell, n0, sumFb):
        self.pos, self.U_, self.UOld_, self.d_, self.cell_, self.n0_, self.sumDeltaFb_ = pos, U, UOld, d, cell, n0, sumFb
        self.vol_ = ns["Vol"](d)
        self.m_ = ns["mass"](rho, d)

    def position(self): return self.pos
    def U(self): return self.U_
    def UOld(self): return self.UOld_
    def d(self): return self.d_
    def cell(self): return self.cell_
    def Vol(self): return self.vol_
    def m(self): return self.m_
    def n0(self): return self.n0_
    def sumDeltaFb(self): return self.sumDeltaFb_


def foam_dialect(src):
    src = re.sub(r"\b(?:Info|Pout)\s*<<[^;]*;", "", src)
    src = re.sub(r"\breduce\s*\([^;]*;", "", src)
    src = re.sub(r"for\s*\(\s*softParticleCloud::iterator\s+pIter\s*=\s*softParticleCloud::begin\(\)\s*;\s*pIter\s*!=\s*"
                 r"softParticleCloud::end\(\)\s*;\s*\+\+pIter\s*,\s*\+\+particleI\s*\)", "forAll(particles_, particleI)", src)
    src = re.sub(r"forAllIter\s*\(\s*softParticleCloud\s*,\s*\*this\s*,\s*iter\s*\)", "forAll(particles_, particleI)", src)
    src = re.sub(r"\b(?:pIter|iter)\(\)", "particles_[particleI]", src)
    src = re.sub(r"\+\+pIter\s*;", "", src)
    src = src.replace(".internalField()", ".f")
    src = src.replace("constant::mathematical::pi", "M_PI_")
    src = src.replace("vector::zero", "Vec(0.0, 0.0, 0.0)")
    src = re.sub(r"\bvector\s*\(", "Vec(", src)
    src = re.sub(r"(\w+)\.(n0|sumDeltaFb)\(\)\s*=(?!=)", r"\1.\2_ =", src)
    return src


def make_value_function(name, args, path, first, last, ns, dialect=None):
    body = translate(path, first, last, dialect)
    src = ("def %s(%s):\n    _result = None\n" % (name, ", ".join(args)) + "\n".join("    " + l for l in body.split("\n"))
           + "\n    return _result\n")
    run(src, ns)


FOAM = dict(MATH, Vec=Vec, mag=fmag, sqr=lambda a: a * a, M_PI_=math.pi, ROOTVSMALL=1.0e-150, debug=0, pIter=0)   # [3P] doubleScalar.H


def cloud_namespace(addParticleOption=0, ecc=(0.0, 0.0, 0.0)):
    ns = dict(FOAM)
    make_value_function("Vol", ["d_"], "lammpsFoam/softParticle.H", 272, 272, ns, foam_dialect)
    # softParticle.C:72  mass_ = density_*4./3.*pi*d_*d_*d_/8.;
    run("def mass(density_, d_):\n    " + translate("lammpsFoam/softParticle.C", 72, 72, foam_dialect).strip()
        + "\n    return mass_\n", ns)
    make_value_function("g1n", ["n"], "lammpsFoam/enhancedCloud.C", 1374, 1383, ns, foam_dialect)
    ns["addParticleOption_"] = addParticleOption
    ns["addParticleBoxEccentricity_"] = Vec(*ecc)
    make_value_function("pointInRegion", ["point", "box"], "lammpsFoam/softParticleCloud.C", 1356, 1416, ns, foam_dialect)
    return ns


def vecs(rows):
    return [Vec(*r) for r in rows]


def vout(vs):
    return [hexs(v.a) for v in vs]


def case_cloud(seed, n, mesh_n, flags, model, n_steps=1, inlet=None, big_accel=False):
    """one mesh block of mesh_n cells with n particles inside; per CFD step: particleToEulerianField (no smoothing) ->
    updateParticleAlpha / Ur -> Jd (the drag model's own lines) -> the force assembly -> calcTcFields (dragSmooth off).
    n_steps > 1 walks the history force through its window: the particle velocities of every step are inputs."""
    rng = random.Random(seed)
    nx, ny, nz = mesh_n
    dx = 3.0e-3
    ncells = nx * ny * nz
    d = [1.0e-3 * rng.uniform(0.5, 1.2) for _ in range(n)]
    rho = 2650.0
    pos = [[rng.uniform(0.02, nx - 0.02) * dx, rng.uniform(0.02, ny - 0.02) * dx, rng.uniform(0.02, nz - 0.02) * dx]
           for _ in range(n)]
    for k in range(0, n, 7):   # some grains within the lubrication window above the y = 0 wall (0.0001 d .. 0.1 d)
        pos[k][1] = 0.5 * d[k] + d[k] * 10.0 ** rng.uniform(-3.5, -1.2)
    cell = [int(p[0] / dx) + nx * (int(p[1] / dx) + ny * int(p[2] / dx)) for p in pos]
    Useq = [[[rng.uniform(-0.3, 0.3) for _ in range(3)] for _ in range(n)] for _ in range(n_steps + 1)]
    if big_accel:              # |DDtUf - dupdt| > 10: the added-mass cap
        for k in range(0, n, 3):
            Useq[1][k] = [u + 2.0e-3 * rng.uniform(-1, 1) for u in Useq[0][k]]
    Uf = [[0.05 * rng.uniform(-1, 1), 0.05 + 0.05 * rng.uniform(-1, 1), 0.05 * rng.uniform(-1, 1)] for _ in range(ncells)]
    UfOld = [[u + 0.01 * rng.uniform(-1, 1) for u in row] for row in Uf]
    DDtUf = [[rng.gauss(0.0, 3.0) for _ in range(3)] for _ in range(ncells)]
    gradp = [[rng.gauss(0.0, 50.0), -9810.0 + rng.gauss(0.0, 50.0), rng.gauss(0.0, 50.0)] for _ in range(ncells)]
    curlU = [[rng.gauss(0.0, 5.0) for _ in range(3)] for _ in range(ncells)]
    V = [dx * dx * dx] * ncells
    deltaT = 5.0e-5
    inp = dict(n=n, mesh_n=list(mesh_n), dx=dx, d=d, rho=rho, pos=pos, cell=cell, U=Useq, Uf=Uf, UfOld=UfOld, DDtUf=DDtUf,
               gradp=gradp, curlU=curlU, deltaT=deltaT, flags=flags, model=model, rhob=1000.0, nub=1.0e-6,
               gravity=[0.0, -9.81, 0.0], inlet=inlet, n_steps=n_steps)
    ns = cloud_namespace(inlet["addParticleOption"] if inlet else 0, inlet["eccentricity"] if inlet else (0.0, 0.0, 0.0))
    jd_path, jd_first, jd_last, jd_result = {0: ("lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C", 104, 132, "tKWenYu"),
                                             1: ("lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C", 105, 143, "_result")}[model]
    jd_code = translate(jd_path, jd_first, jd_last)

    class Drag:
        def Jd(self, Ur):
            out = []
            for k in range(len(Ur)):   # the field algebra of the model, one element at a time (as case_drag)
                dns = dict(MATH, Ur=Field([Ur[k]]), alpha_=Field([ns["pAlpha_"][k]]), pd_=Field([ns["pDia_"][k]]),
                           nuf_=inp["nub"], rhof_=inp["rhob"], pow=fpow, max=fmax, sqrt=fsqrt, sqr=fsqr, ROOTVSMALL=1.0e-150,
                           scalar=float)
                run(jd_code, dns)
                out.append(dns[jd_result][0])
            return Field(out)

    RT = type("RT", (), {})()
    ns.update(drag_=Drag(), runTime=lambda: RT, mesh_=type("Mesh", (), {"V": lambda self: Field(V)})(),
              gravity_=Vec(*inp["gravity"]), rhob_=inp["rhob"], nub_=inp["nub"],
              particleDragFlag_=flags.get("particleDrag", 1), particlePressureGradFlag_=flags.get("particlePressureGrad", 1),
              particleBuoyancyFlag_=flags.get("particleBuoyancy", 0), particleAddedMassFlag_=flags.get("particleAddedMass", 0),
              particleLiftForceFlag_=flags.get("particleLift", 0), particleHistoryForceFlag_=flags.get("particleHistoryForce", 0),
              lubricationFlag_=flags.get("lubricationForce", 0),
              inletForceRatio_=Vec(*(inlet["inletForce"] if inlet else (0.0, 0.0, 0.0))),
              inletBox_=Tensor9(inlet["inletBox"] if inlet else [0.0] * 9),
              alphaSmoothFlag_=0, UpSmoothFlag_=0, dragSmoothFlag_=0, semiImplicit=0)
    code_scatter = translate("lammpsFoam/enhancedCloud.C", 913, 979, foam_dialect)
    code_alpha = translate("lammpsFoam/enhancedCloud.C", 59, 75, foam_dialect)
    code_ur = translate("lammpsFoam/enhancedCloud.C", 88, 108, foam_dialect)
    code_dia = translate("lammpsFoam/enhancedCloud.C", 41, 51, foam_dialect)
    code_drag = translate("lammpsFoam/enhancedCloud.C", 129, 311, foam_dialect)
    code_tc = translate("lammpsFoam/enhancedCloud.C", 318, 439, foam_dialect)
    if "--show" in sys.argv:
        print(code_scatter, code_ur, code_drag, code_tc, sep="\n# ----\n")
    parts = [Particle(ns, Vec(*pos[k]), Vec(*Useq[0][k]), Vec(*Useq[0][k]), d[k], rho, cell[k], 0.0, Vec()) for k in range(n)]
    ns["particles_"] = parts
    ns["particleCount_"] = n
    inp["mass"] = [p.m_ for p in parts]   # (softParticle.C:72, executed: what the inlet override multiplies by)
    # (calcTcFields calls the three list updates itself, :323-325)
    ns.update(updateParticleAlpha=lambda: run(code_alpha, ns), updateParticleUr=lambda: run(code_ur, ns),
              setupParticleDia=lambda: run(code_dia, ns))
    steps = []
    for step in range(1, n_steps + 1):
        for k, p in enumerate(parts):
            p.UOld_, p.U_ = p.U_, Vec(*Useq[step][k])
        RT.deltaT = lambda: type("DT", (), {"value": lambda self: deltaT})()
        RT.timeIndex = lambda step=step: step
        ns.update(gamma_=GeoField([0.0] * ncells), Ue_=GeoField([Vec() for _ in range(ncells)]),
                  UfSmoothed_=GeoField(vecs(Uf), old=GeoField(vecs(UfOld))), DDtUf_=GeoField(vecs(DDtUf)),
                  gradp=Field(vecs(gradp)), curlU=Field(vecs(curlU)), Omega_=GeoField([0.0] * ncells),
                  Asrc_=GeoField([Vec() for _ in range(ncells)]), Asrc2_=GeoField([Vec() for _ in range(ncells)]),
                  pDia_=Field([0.0] * n), pAlpha_=Field([0.0] * n), Uri_=Field([Vec()] * n), magUri_=Field([0.0] * n),
                  pDrag_=Field([Vec()] * n), pDuDt_=Field([Vec()] * n), Jd_=Field([0.0] * n))
        for name in ("pDia_", "pAlpha_", "Uri_", "magUri_", "pDrag_", "pDuDt_"):
            ns[name].setSize = lambda m: None
        run(code_scatter, ns)                       # gamma_, Ue_
        gamma_now, Ue_now = list(ns["gamma_"].f), list(ns["Ue_"].f)
        run(code_dia, ns); run(code_alpha, ns); run(code_ur, ns)
        run(code_drag, ns)                          # Jd_, pDrag_, pDuDt_, history state
        out = dict(gamma=hexs(gamma_now), Ue=vout(Ue_now), Uri=vout(ns["Uri_"]), magUri=hexs(ns["magUri_"]),
                   Jd=hexs(ns["Jd_"]), pDrag=vout(ns["pDrag_"]), pDuDt=vout(ns["pDuDt_"]),
                   sumDeltaFb=vout([p.sumDeltaFb_ for p in parts]), n0=hexs([p.n0_ for p in parts]))
        # liftDragCoeffs.H:6-14 caps alpha before calcTcFields; the case stays below the cap, the field goes in as it is
        run(code_tc, ns)
        out.update(Asrc=vout(ns["Asrc_"].f), Omega=hexs(ns["Omega_"].f), step=step)
        steps.append(out)
    if n_steps > 3:   # a long walk: the first two steps, the first one after a history-window reset, the last two
        reset = next((k for k, o in enumerate(steps) if any(float.fromhex(v) > 0.0 for v in o["n0"])), n_steps - 2)
        keep = sorted({0, 1, reset, min(reset + 1, n_steps - 1), n_steps - 2, n_steps - 1})
        steps = [steps[k] for k in keep]
    return dict(inp=inp, out=steps)


def main():
    pins = {"_about": "numbers computed by the reference's own source lines (tests/golden/make_reference_pins.py); "
                      "floats of the outputs in C99 hex notation"}
    code = translate("interfaceToLammps/pair_gran_hertzFix_history.cpp", 109, 286)
    if "--show" in sys.argv:
        print(code)
    pins["pair_gran_hertzFix_history.cpp:109-286"] = [
        case_hertz(code, 11, 27, 27, False, 1, False), case_hertz(code, 12, 48, 30, True, 1, False),
        case_hertz(code, 13, 48, 36, True, 0, True), case_hertz(code, 14, 64, 64, True, 1, True),
        case_hertz(code, 15, 48, 36, True, 1, True, rigid=True)]
    code = translate("interfaceToLammps/fix_cohesive.cpp", 161, 262)
    if "--show" in sys.argv:
        print(code)
    pins["fix_cohesive.cpp:161-262"] = [
        case_cohesive(code, 21, 40, 40, 0, 0, 1.0e-9, False), case_cohesive(code, 22, 48, 30, 0, 0, 4.0e-9, True),
        case_cohesive(code, 23, 48, 30, 1, 0, 1.0e-9, True), case_cohesive(code, 24, 40, 24, 1, 1, 4.0e-9, False),
        case_cohesive(code, 25, 40, 24, 0, 1, 1.0e-9, False)]
    code = translate("interfaceToLammps/fix_fluid_drag.cpp", 143, 163)
    if "--show" in sys.argv:
        print(code)
    pins["fix_fluid_drag.cpp:143-163"] = [case_fdrag(code, 31, 50, 0.0), case_fdrag(code, 32, 50, 1000.0)]
    pins["fix_wall_granFix.cpp:286-344,361-436,446-553,563-678"] = [
        case_wall(61, 60, 2, 1, 1), case_wall(62, 60, 2, 0, 0), case_wall(63, 60, 1, 2, 1), case_wall(64, 60, 0, 1, 1)]
    pins["pair_lubricate_poly.cpp:193-407,539-559"] = [
        case_lubricate(71, 64, 64, 1, 0, 1), case_lubricate(72, 64, 40, 1, 1, 1), case_lubricate(73, 48, 48, 0, 1, 0),
        case_lubricate(74, 48, 30, 0, 0, 1), case_lubricate(75, 64, 64, 1, 0, 1, cut_inner_d=0.8)]
    code = translate("lammpsFoam/dragModels/ErgunWenYu/ErgunWenYu.C", 104, 132)
    if "--show" in sys.argv:
        print(code)
    pins["ErgunWenYu.C:104-132"] = [case_drag(code, "tKWenYu", 41, 200, 1.0e-6, 1000.0),
                                    case_drag(code, "tKWenYu", 42, 100, 1.5e-5, 1.2)]
    code = translate("lammpsFoam/dragModels/SyamlalOBrien/SyamlalOBrien.C", 105, 143)
    if "--show" in sys.argv:
        print(code)
    pins["SyamlalOBrien.C:105-143"] = [case_drag(code, "_result", 51, 200, 1.0e-6, 1000.0),
                                       case_drag(code, "_result", 52, 100, 1.5e-5, 1.2)]
    allf = dict(particleDrag=1, particlePressureGrad=1, particleBuoyancy=1, particleAddedMass=1, particleLift=1,
                lubricationForce=1)
    pins["enhancedCloud.C:41-108,129-311,318-439,913-979"] = [
        case_cloud(81, 160, (3, 4, 3), dict(particleDrag=1, particlePressureGrad=1), 0),
        case_cloud(82, 200, (4, 3, 3), allf, 0, big_accel=True),
        case_cloud(83, 120, (3, 3, 3), allf, 1, n_steps=2, big_accel=True),
        case_cloud(84, 40, (2, 3, 2), dict(allf, particleHistoryForce=1), 0, n_steps=260),
        case_cloud(85, 150, (3, 4, 3), dict(particleDrag=1, particlePressureGrad=1), 0,
                   inlet=dict(addParticleOption=1, inletForce=[0.3, 0.0, 0.0], inletBox=[0.0, 4.5e-3, 0.0, 6.0e-3, 0.0, 9.0e-3, 0.0, 0.0, 0.0],
                              eccentricity=[0.0, 0.0, 0.0])),
        case_cloud(86, 150, (3, 4, 3), dict(particleDrag=1, particlePressureGrad=1), 0,
                   inlet=dict(addParticleOption=2, inletForce=[0.0, 0.2, 0.0], inletBox=[4.5e-3, 4.5e-3, 0.0, 12.0e-3, 4.5e-3, 4.5e-3, 1.0e-3, 4.0e-3, 0.0],
                              eccentricity=[3.0e-4, 0.0, -2.0e-4]))]
    with open(os.path.join(HERE, "reference_pins.json"), "w") as fh:
        json.dump(pins, fh, separators=(",", ":"))
    print("wrote reference_pins.json: " + ", ".join("%s x%d" % (k, len(v)) for k, v in pins.items() if k[0] != "_"))


if __name__ == "__main__":
    main()
