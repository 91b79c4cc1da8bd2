/* declarations-only MPI stub: see README.md (syntax check of the boundary headers, nothing else) */
#ifndef PIB_STUB_MPI_H
#define PIB_STUB_MPI_H
typedef struct pib_stub_mpi_comm *MPI_Comm;
typedef struct pib_stub_mpi_info *MPI_Info;
typedef int MPI_Datatype;
#define MPI_COMM_NULL ((MPI_Comm)0)
#define MPI_INFO_NULL ((MPI_Info)0)
#define MPI_BYTE ((MPI_Datatype)1)
#define MPI_COMM_TYPE_SHARED 1
#ifdef __cplusplus
extern "C" {
#endif
int MPI_Comm_rank(MPI_Comm comm, int *rank);
int MPI_Comm_size(MPI_Comm comm, int *size);
int MPI_Bcast(void *buffer, int count, MPI_Datatype datatype, int root, MPI_Comm comm);
int MPI_Comm_split_type(MPI_Comm comm, int split_type, int key, MPI_Info info, MPI_Comm *newcomm);
int MPI_Comm_free(MPI_Comm *comm);
#ifdef __cplusplus
}
#endif
#endif
