# Parity of the libgdca.so binding with the reference on the reference's own test data, for a machine that has Julia,
# GaussDCA.jl, DCAUtils.jl and an MI355X (the build image has no Julia: this file has never been executed there).
#   LIBGDCA=/path/to/libgdca.so julia --project=julia -e 'using Pkg; Pkg.test()'
# Mirrors test/runtests.jl of GaussDCA.jl (the compare helper and the four golden cases, test/runtests.jl:29-86):
# identical (i, j) keys, scores within 1e-6 relative of the reference's own gDCA on the same file.
using Test
import GaussDCA
import GaussDCAHip

const datadir = joinpath(dirname(pathof(GaussDCA)), "..", "test", "data")

function agree(R1, R2; rtol = 1e-6)
    length(R1) == length(R2) || return false
    d2 = Dict((i, j) => x for (i, j, x) in R2)
    all(haskey(d2, (i, j)) && isapprox(x, d2[(i, j)]; rtol = rtol, atol = 1e-9) for (i, j, x) in R1)
end

@testset "libgdca.so vs GaussDCA.jl on test/data" begin
    for (file, kw) in (("small.fasta.gz", (;)),
                       ("small.fasta.gz", (pseudocount = 0.2, score = :DI, remove_dups = true)),
                       ("small.fasta.gz", (pseudocount = 0.2, score = :DI, min_separation = 1, θ = 0.3)),
                       ("large.fasta.gz", (pseudocount = 0.2, score = :DI, remove_dups = true)))
        f = joinpath(datadir, file)
        @test agree(GaussDCAHip.gDCA(f; kw...), GaussDCA.gDCA(f; kw...))
    end
end
